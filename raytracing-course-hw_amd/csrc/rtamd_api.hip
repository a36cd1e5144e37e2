// C-ABI of the render path (include/rtamd.h): scene upload, render launch, output handling.
// No CPU fallback exists: every entry point that needs the GPU fails with RT_ERR_NO_DEVICE / RT_ERR_HIP
// when HIP is unusable.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/rtamd.h"
#include "host/host_scene.h"
#include "host/png.h"
#include "host/scene_prep.h"
#include "host/shared_prep.h"
#include "host/hip_check.h"
#include "device/rt_node_grid.h"
#include "host/device_build.h"
#include "device/rt_kernels_hw8.h"
#include "device/rt_wavefront.h"
#include "device/rt_persistent.h"
#include "device/rt_kernels_hw6.h"
#include "device/rt_persistent_hw6.h"
#include "device/rt_kernels_txt.h"
#include "device/rt_kernels_hw2.h"
#include "device/rt_kernels_hw4.h"
#include "device/rt_kernels_hw5.h"
#include <cstdlib>

namespace rtamd {
static thread_local std::string g_last_error;
void set_error(const std::string &msg) { g_last_error = msg; }
}
using namespace rtamd;

namespace {

template <class T> T *upload(const std::vector<T> &v, uint64_t &bytes) {
    if (v.empty()) { // keep pointers valid: one dummy element
        void *p = nullptr;
        HIP_CHECK(hipMalloc(&p, sizeof(T) > 16 ? sizeof(T) : 16));
        HIP_CHECK(hipMemset(p, 0, sizeof(T) > 16 ? sizeof(T) : 16));
        return (T *)p;
    }
    void *p = nullptr;
    size_t n = v.size() * sizeof(T);
    HIP_CHECK(hipMalloc(&p, n));
    HIP_CHECK(hipMemcpy(p, v.data(), n, hipMemcpyHostToDevice));
    bytes += n;
    return (T *)p;
}

double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Device buffer owned by one rt_render call (host-output renders): freed on every way out of the call.
struct OwnedDev {
    void *p = nullptr;
    ~OwnedDev() { if (p) (void)hipFree(p); }
};

int fail(int code, const std::string &msg) {
    set_error(msg);
    return code;
}

} // namespace

struct rt_scene {
    int device = 0;
    SceneView view{};
    SceneView6 view6{};
    SceneViewTxt viewt{};
    SceneView5 view5{};
    bool txt_has_triangles = false; // TRIANGLE figures exist only in the hw5 grammar: such a scene renders with RT_INTEGRATOR_HW5 only
    int flavor = RT_INTEGRATOR_HW8; // which integrator this scene was prepared for
    bool hw6_lds_stack = false, hw6_pt_stack = false;
    uint32_t light_walk_depth = 0;   // hw8: depth of the tree the persistent kernel's light walker uses
    uint32_t wide_depth = 0, wide_light_depth = 0; // levels of the four-wide forms of the two walk trees (rt_types.h GpuNode4Q)
    std::vector<void *> allocations;
    rt_scene_info info{};
    std::vector<uint32_t> light_order;
    float fov_y = 0;
    uint32_t *d_work_counter = nullptr;
    unsigned long long *d_counters = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    int n_cus = 256;
    // wavefront path state (grown on demand, reused across renders)
    dev::WfView wf{};
    size_t wf_slots = 0, wf_levels = 0, wf_ctr_words = 0, wf_ovf_words = 0;
    int wf_pipes = 1;                // pipelines of the last wavefront render and the counter words of each
    size_t wf_ctr_block = 0;
    hipStream_t wf_streams[4] = {nullptr, nullptr, nullptr, nullptr}; // one per pipeline when a render uses more than one
    hipEvent_t ev_fork = nullptr, ev_join[4] = {nullptr, nullptr, nullptr, nullptr};
    std::vector<void *> wf_allocs;
    float *d_partial = nullptr;      // throughput mode: per-stream pixel sums
    size_t partial_bytes = 0;
    std::vector<hipEvent_t> ev_pool; // brackets every launch of the dominant kernel when stats are requested
    bool device_tree = false;                 // RT_BUILD_DEVICE_BVH: figure order = LOAD order, no reference trees
    unsigned long long *d_pt_debug = nullptr; // persistent pipeline: per workgroup {start, exit time, paths} (RTAMD_DEBUG_COUNTERS)
    float4 *pt_r0 = nullptr;         // persistent pipeline: path records of one pass
    uint32_t *pt_groups = nullptr;   // [cost per group | group_ofs (n_blocks + 1) | group_ids]: the re-deal between the phases of a frame
    size_t pt_slots = 0, pt_levels = 0, pt_group_words = 0;
    uint32_t pt_passes = 0, pt_blocks = 0, pt_launches = 0;
    double pt_rebalance_ms = 0, pt_imbalance = 0;
    float4 *pt6_r0 = nullptr;        // hw6 persistent pipeline: path records (frames included) of one pass
    size_t pt6_slots = 0;
    int pipeline = 0;                // RT_PIPELINE_* of the last render
    void free_wf() {
        for (void *p : wf_allocs) (void)hipFree(p);
        wf_allocs.clear();
        wf = dev::WfView{};
        wf_slots = wf_levels = wf_ctr_words = wf_ovf_words = 0;
    }
    ~rt_scene() {
        free_wf();
        if (d_partial) (void)hipFree(d_partial);
        if (pt_r0) (void)hipFree(pt_r0);
        if (pt_groups) (void)hipFree(pt_groups);
        if (pt6_r0) (void)hipFree(pt6_r0);
        if (d_pt_debug) (void)hipFree(d_pt_debug);
        for (void *p : allocations) (void)hipFree(p);
        if (ev_start) (void)hipEventDestroy(ev_start);
        if (ev_stop) (void)hipEventDestroy(ev_stop);
        for (hipEvent_t e : ev_pool) (void)hipEventDestroy(e);
        for (int h = 0; h < 4; h++) { if (wf_streams[h]) (void)hipStreamDestroy(wf_streams[h]); if (ev_join[h]) (void)hipEventDestroy(ev_join[h]); }
        if (ev_fork) (void)hipEventDestroy(ev_fork);
    }
};

extern "C" {

int rt_abi_version(void) { return RTAMD_ABI_VERSION; }
const char *rt_last_error(void) { return g_last_error.c_str(); }

int rt_scene_create(const rt_scene_desc *desc, rt_scene **out) { return rtamd::scene_create_shared(desc, out, nullptr); }

} // extern "C"

// rt_scene_create with the host-side preparation of an hw8 / hw7 scene (the replay of the reference's figure and light order:
// ~0.4 s for the benchmark scene) optionally taken from `shared`: the first caller fills it, the others wait for it and only upload.
// rt_multi_create gives all its devices the same one (rtamd_multi.hip); the plain C entry point passes none.
int rtamd::scene_create_shared(const rt_scene_desc *desc, rt_scene **out, rtamd::SharedPrep *shared) {
    if (!desc || !out) return fail(RT_ERR_INVALID_ARG, "rt_scene_create: null argument");
    if (desc->struct_size != sizeof(rt_scene_desc)) return fail(RT_ERR_INVALID_ARG, "rt_scene_create: struct_size mismatch (ABI skew)");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(RT_ERR_NO_DEVICE, "rt_scene_create: no HIP device available (this library has no CPU fallback)");
    try {
        std::unique_ptr<rt_scene> s(new rt_scene());
        HIP_CHECK(hipGetDevice(&s->device));
        hipDeviceProp_t prop;
        HIP_CHECK(hipGetDeviceProperties(&prop, s->device));
        s->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        double t0 = now_ms();
        // Analytic primitives only: a .txt scene (hw1 / hw3).
        // (A scene without any figure counts as a .txt scene when its camera carries only CAMERA_FOV_X, as the .txt loader leaves it.)
        if (desc->n_triangles == 0 && (desc->n_primitives > 0 || (desc->camera.fov_x != 0.f && desc->camera.fov_y == 0.f))) {
            std::vector<GpuPrim> prims(desc->n_primitives);
            for (uint32_t i = 0; i < desc->n_primitives; i++) {
                const rt_primitive &p = desc->primitives[i];
                GpuPrim &g = prims[i];
                memset(&g, 0, sizeof g);
                if (p.type < RT_PRIM_ELLIPSOID || p.type > RT_PRIM_TRIANGLE) return fail(RT_ERR_INVALID_ARG, "rt_scene_create: bad primitive type");
                if (p.type == RT_PRIM_TRIANGLE) s->txt_has_triangles = true;
                for (int k = 0; k < 3; k++) { g.data[k] = p.data[k]; g.position[k] = p.position[k]; g.color[k] = p.color[k]; g.emission[k] = p.emission[k]; }
                for (int k = 0; k < 4; k++) g.rotation[k] = p.rotation[k];
                g.type = p.type; g.kind = p.kind; g.ior = p.ior;
            }
            uint64_t bytes = 0;
            SceneViewTxt &V = s->viewt;
            V.prims = upload(prims, bytes);
            s->allocations.push_back((void *)V.prims);
            V.n_prims = desc->n_primitives;
            for (int k = 0; k < 3; k++) {
                V.cam_pos[k] = desc->camera.position[k]; V.cam_right[k] = desc->camera.right[k];
                V.cam_up[k] = desc->camera.up[k]; V.cam_fwd[k] = desc->camera.forward[k];
                V.bg[k] = desc->bg_color[k];
            }
            V.tan_fov_x = (float)std::tan((double)(desc->camera.fov_x / 2)); // hw3/src/scene.cpp:100
            V.tan_fov_x_f = tanf(desc->camera.fov_x / 2);                    // hw1/src/scene.cpp:23, hw2/src/scene.cpp:91 (<math.h>: float overload)
            std::vector<GpuLight> lights(desc->n_lights);
            for (uint32_t i = 0; i < desc->n_lights; i++) {
                const rt_light &L = desc->lights[i];
                GpuLight &g = lights[i];
                memset(&g, 0, sizeof g);
                if (L.type != RT_LIGHT_POINT && L.type != RT_LIGHT_DIRECTIONAL) return fail(RT_ERR_INVALID_ARG, "rt_scene_create: bad light type");
                g.type = L.type;
                for (int k = 0; k < 3; k++) { g.intensity[k] = L.intensity[k]; g.position[k] = L.position[k]; g.attenuation[k] = L.attenuation[k]; g.direction[k] = L.direction[k]; }
            }
            V.n_lights = desc->n_lights;
            if (desc->n_lights) { V.lights = upload(lights, bytes); s->allocations.push_back((void *)V.lights); }
            for (int k = 0; k < 3; k++) V.ambient[k] = desc->ambient_light[k];
            std::vector<uint32_t> light_prims; // hw4/src/scene.cpp:12-21
            for (uint32_t i = 0; i < desc->n_primitives; i++) {
                const rt_primitive &p = desc->primitives[i];
                if ((p.emission[0] > 0 || p.emission[1] > 0 || p.emission[2] > 0) && (p.type == RT_PRIM_BOX || p.type == RT_PRIM_ELLIPSOID)) light_prims.push_back(i);
            }
            V.n_light_prims = (uint32_t)light_prims.size();
            if (!light_prims.empty()) { V.light_prims = upload(light_prims, bytes); s->allocations.push_back((void *)V.light_prims); }
            { // hw5 structures: reference figure order, BVH over the non-planes, light list + light BVH
                PreparedScene5 P5;
                prepare_scene_hw5(*desc, P5);
                SceneView5 &V5 = s->view5;
                V5.nodes = upload(P5.nodes, bytes); s->allocations.push_back((void *)V5.nodes);
                V5.figs = upload(P5.figs, bytes); s->allocations.push_back((void *)V5.figs);
                V5.light_nodes = upload(P5.light_nodes, bytes); s->allocations.push_back((void *)V5.light_nodes);
                V5.ref_nodes = upload(P5.ref_nodes, bytes); s->allocations.push_back((void *)V5.ref_nodes);
                V5.ref_light_nodes = upload(P5.ref_light_nodes, bytes); s->allocations.push_back((void *)V5.ref_light_nodes);
                if (!P5.lights.empty()) { V5.lights = upload(P5.lights, bytes); s->allocations.push_back((void *)V5.lights); }
                V5.n_figs = (uint32_t)P5.figs.size(); V5.n_nonplanes = P5.n_nonplanes; V5.n_lights = (uint32_t)P5.lights.size();
                for (int k = 0; k < 3; k++) {
                    V5.cam_pos[k] = V.cam_pos[k]; V5.cam_right[k] = V.cam_right[k]; V5.cam_up[k] = V.cam_up[k]; V5.cam_fwd[k] = V.cam_fwd[k]; V5.bg[k] = V.bg[k];
                }
                V5.tan_fov_x = V.tan_fov_x;
                s->light_order = P5.light_order;
                s->info.n_lights = V5.n_lights; s->info.n_bvh_nodes = (uint32_t)P5.nodes.size(); s->info.n_light_bvh_nodes = (uint32_t)P5.light_nodes.size();
                s->info.bvh_depth = P5.bvh_depth; s->info.light_bvh_depth = P5.light_bvh_depth;
            }
            s->flavor = RT_INTEGRATOR_HW3;
            HIP_CHECK(hipMalloc((void **)&s->d_work_counter, 64));
            s->allocations.push_back(s->d_work_counter);
            HIP_CHECK(hipMalloc((void **)&s->d_counters, 512));
            s->allocations.push_back(s->d_counters);
            HIP_CHECK(hipEventCreate(&s->ev_start));
            HIP_CHECK(hipEventCreate(&s->ev_stop));
            HIP_CHECK(hipDeviceSynchronize());
            s->info.device_bytes = bytes;
            s->info.prep_ms = 0; s->info.upload_ms = now_ms() - t0;
            *out = s.release();
            return RT_OK;
        }
        // A scene without per-vertex normals can only be an hw6 scene (flat shading, hw6/src/sceneio.cpp:186-225).
        if (desc->n_triangles && !desc->normals) {
            // hw6's scene tree is the library's own (the reference's is degenerate, rt_kernels_hw6.h), so it can be built on the GPU
            // without touching the replay: the tie rule reads the reference's figure index from the record.
            const bool tree_on_device = desc->n_triangles >= 64 && !getenv("RTAMD_HOST_BVH");
            PreparedScene6 P6;
            prepare_scene_hw6(*desc, P6, tree_on_device);
            double t1 = now_ms();
            uint64_t bytes = 0;
            SceneView6 &V = s->view6;
            auto keep = [&](auto *p) { s->allocations.push_back((void *)p); return p; };
            if (tree_on_device) {
                uint64_t scratch = 0;
                Tri6 *d_load = upload(P6.tris, bytes);
                float *d_boxes = upload(P6.boxes8, scratch);
                DeviceTree t;
                Tri6 *d_tris = nullptr;
                try {
                    t = build_tree_on_device(d_boxes, desc->n_triangles, P6.box_pad, 28); // hw6 walkers: 36-entry stack columns
                    HIP_CHECK(hipMalloc((void **)&d_tris, (size_t)desc->n_triangles * sizeof(Tri6)));
                    gather_records(d_load, d_tris, t, desc->n_triangles, sizeof(Tri6), 13); // word 13 = Tri6::last
                    HIP_CHECK(hipDeviceSynchronize());
                } catch (...) {
                    (void)hipFree(d_load); (void)hipFree(d_boxes); if (d_tris) (void)hipFree(d_tris);
                    free_device_tree(t);
                    throw;
                }
                (void)hipFree(d_load); (void)hipFree(d_boxes);
                (void)hipFree(t.order); (void)hipFree(t.last); t.order = nullptr; t.last = nullptr;
                V.nodes = keep(t.nodes);
                V.tris = keep(d_tris);
                bytes += (uint64_t)t.n_nodes * sizeof(GpuNode);
                P6.bvh_depth = t.depth;
                P6.nodes.resize(t.n_nodes); // node count for rt_scene_info
                s->info.bvh_build_ms = t.build_ms; s->info.bvh_on_device = 1;
            } else {
                V.nodes = keep(upload(P6.nodes, bytes));
                V.tris = keep(upload(P6.tris, bytes));
            }
            if (P6.bvh_depth > RT6_STACK_SIZE - 2 || P6.light_bvh_depth > RT6_STACK_SIZE - 2 || P6.fast_light_bvh_depth > RT6_STACK_SIZE - 2)
                return fail(RT_ERR_LIMIT, "scene BVH deeper than the kernel's traversal stack (" + std::to_string(P6.bvh_depth) + "/" +
                                              std::to_string(P6.light_bvh_depth) + ")");
            s->hw6_lds_stack = P6.bvh_depth <= RT6_LDS_STACK && P6.fast_light_bvh_depth <= RT6_LDS_STACK; // both own trees fit the LDS stack columns
            s->hw6_pt_stack = P6.bvh_depth <= P6_STACK && P6.fast_light_bvh_depth <= P6_STACK;           // ... of the persistent pipeline
            V.light_nodes = keep(upload(P6.light_nodes, bytes));
            V.lights = keep(upload(P6.lights, bytes));
            V.fast_light_nodes = keep(upload(P6.fast_light_nodes, bytes));
            V.fast_lights = keep(upload(P6.fast_lights, bytes));
            V.light_ref = keep(upload(P6.light_ref, bytes));
            V.ref_nodes = keep(upload(P6.ref_nodes, bytes));
            V.ref_light_nodes = keep(upload(P6.ref_light_nodes, bytes));
            V.ref_tris = keep(upload(P6.ref_tris, bytes));
            V.tri_box = keep(upload(P6.tri_box, bytes));
            V.box_c2 = P6.box_c2; V.box_c2x = 1.25f * P6.box_c2;
            V.cull_k = getenv("RTAMD_CULL_K") ? (float)atof(getenv("RTAMD_CULL_K")) : 0.0078125f;
            V.exact_boxes = getenv("RTAMD_NO_EXACT_BOXES") ? 0u : 1u;
            V.light_sep = keep(upload(P6.light_sep, bytes));
            V.materials = keep(upload(P6.materials, bytes));
            V.n_tris = desc->n_triangles;
            V.n_lights = (uint32_t)P6.lights.size();
            V.n_components = P6.lights.empty() ? 1u : 2u; // hw6/src/scene.cpp:8-16
            V.n_lights_f = (float)V.n_lights; V.n_components_f = (float)V.n_components;
            for (int k = 0; k < 3; k++) {
                V.cam_pos[k] = desc->camera.position[k]; V.cam_right[k] = desc->camera.right[k];
                V.cam_up[k] = desc->camera.up[k]; V.cam_fwd[k] = desc->camera.forward[k];
                V.bg[k] = desc->bg_color[k];
            }
            V.tan_fov_y = (float)std::tan((double)(desc->camera.fov_y / 2));
            s->view.tan_fov_y = V.tan_fov_y;
            s->flavor = RT_INTEGRATOR_HW6;
            {   // the walk nodes of the persistent pipeline (rt_types.h GpuNode4Q), as for hw8
                float glo[3], ghi[3];
                for (int k = 0; k < 3; k++) glo[k] = ghi[k] = V.cam_pos[k];
                join_root_box(V.nodes, glo, ghi);
                join_root_box(V.fast_light_nodes, glo, ghi);
                V.grid = make_node_grid(glo, ghi);
                uint32_t n4 = 0, n4l = 0, d4 = 0, d4l = 0;
                V.nodes4 = keep(widen_nodes(V.nodes, (uint32_t)P6.nodes.size(), V.grid, n4, d4));
                V.fast_light_nodes4 = keep(widen_nodes(V.fast_light_nodes, (uint32_t)P6.fast_light_nodes.size(), V.grid, n4l, d4l));
                bytes += ((uint64_t)n4 + n4l) * sizeof(GpuNode4Q);
                s->wide_depth = d4; s->wide_light_depth = d4l;
            }
            HIP_CHECK(hipMalloc((void **)&s->d_work_counter, 64));
            s->allocations.push_back(s->d_work_counter);
            HIP_CHECK(hipMalloc((void **)&s->d_counters, 512));
            s->allocations.push_back(s->d_counters);
            HIP_CHECK(hipEventCreate(&s->ev_start));
            HIP_CHECK(hipEventCreate(&s->ev_stop));
            HIP_CHECK(hipDeviceSynchronize());
            double t2 = now_ms();
            s->light_order = P6.light_order;
            s->info.n_triangles = desc->n_triangles; s->info.n_lights = V.n_lights;
            s->info.n_bvh_nodes = (uint32_t)P6.nodes.size(); s->info.n_light_bvh_nodes = (uint32_t)P6.light_nodes.size();
            s->info.bvh_depth = P6.bvh_depth; s->info.light_bvh_depth = P6.light_bvh_depth;
            s->info.device_bytes = bytes; s->info.prep_ms = t1 - t0; s->info.upload_ms = t2 - t1;
            *out = s.release();
            return RT_OK;
        }
        if (desc->build_flags & ~RT_BUILD_DEVICE_BVH) return fail(RT_ERR_INVALID_ARG, "rt_scene_create: unknown build_flags");
        // Two things a scene tree is needed for.  The replay needs the reference's FIGURE ORDER (tie rule, light numbering) and, for the
        // rare hits at a box boundary, the reference's own tree (exact walks): the host replays the reference's builder for those
        // (prepare_scene) unless RT_BUILD_DEVICE_BVH gives the order up.  The walkers need a good tree of bounded depth, and a closest
        // hit does not depend on which: that one is built on the GPU (device/rt_bvh_build.h) over the records in figure order, each of
        // which carries its figure index (RTAMD_HOST_BVH=1, or a handful of triangles: the walkers use the reference topology).
        const bool fast_build = (desc->build_flags & RT_BUILD_DEVICE_BVH) && desc->n_triangles >= 64;
        const bool walk_tree_on_device = desc->n_triangles >= 64 && (fast_build || !getenv("RTAMD_HOST_BVH"));
        PreparedScene P_local;
        if (shared) std::call_once(shared->once, [&] { try { prepare_scene(*desc, shared->P, fast_build); } catch (...) { shared->error = std::current_exception(); } });
        else prepare_scene(*desc, P_local, fast_build);
        if (shared && shared->error) std::rethrow_exception(shared->error);
        const PreparedScene &P = shared ? shared->P : P_local; // read-only from here on (several devices may be uploading from it)
        uint32_t bvh_depth = P.bvh_depth, n_nodes = (uint32_t)P.nodes.size();
        double t1 = now_ms();
        uint64_t bytes = 0;
        SceneView &V = s->view;
        auto keep = [&](auto *p) { s->allocations.push_back((void *)p); return p; };
        V.tri_isect = keep(upload(P.isect, bytes));
        V.tri_shade = keep(upload(P.shade, bytes));
        V.tri_box = keep(upload(P.tri_box, bytes));
        uint32_t ref_depth = P.bvh_depth;
        if (walk_tree_on_device) {
            const uint32_t n = desc->n_triangles;
            // The persistent kernel's walkers take two levels per step (GpuNode4Q) and hold up to three entries per step in a stack column
            // of P8_STACK entries; a walk that runs out of room is redone by the exact role with a stack of its own (rt_persistent.h).
            const uint32_t walk_depth_cap = P8_STACK;
            DeviceTree t;
            TriIsect *d_walk = nullptr;
            try {
                if (!P.walk_box.empty()) { // reference leaf boxes (scene_prep.h)
                    uint64_t scratch = 0;
                    float *d_walk_box = upload(P.walk_box, scratch);
                    try { t = build_tree_on_device(d_walk_box, n, P.box_pad, walk_depth_cap); } catch (...) { (void)hipFree(d_walk_box); throw; }
                    (void)hipFree(d_walk_box);
                } else t = build_tree_on_device(V.tri_box, n, P.box_pad, walk_depth_cap);
                HIP_CHECK(hipMalloc((void **)&d_walk, (size_t)n * sizeof(TriIsect)));
                gather_records(V.tri_isect, d_walk, t, n, sizeof(TriIsect), 11, true); // word 11 = TriIsect::pad: figure index << 1 | leaf mark
                HIP_CHECK(hipDeviceSynchronize());
            } catch (...) {
                if (d_walk) (void)hipFree(d_walk);
                free_device_tree(t);
                throw;
            }
            (void)hipFree(t.order); (void)hipFree(t.last); t.order = nullptr; t.last = nullptr;
            V.nodes = keep(t.nodes); V.tri_walk = keep(d_walk);
            bytes += (uint64_t)t.n_nodes * sizeof(GpuNode) + (uint64_t)n * sizeof(TriIsect);
            bvh_depth = t.depth;
            n_nodes = t.n_nodes;
            s->info.bvh_build_ms = t.build_ms; s->info.bvh_on_device = 1;
            s->device_tree = fast_build;
        } else {
            V.nodes = keep(upload(P.nodes, bytes));
            V.tri_walk = V.tri_isect;
        }
        if (ref_depth > RT_STACK_SIZE - 2) // the exact walks keep a private stack over the reference's own tree
            return fail(RT_ERR_LIMIT, "the reference's scene BVH is deeper than the exact walk's stack (" + std::to_string(ref_depth) + ")");
        if (bvh_depth > RT_STACK_SIZE - 2 || P.light_bvh_depth > RT_STACK_SIZE - 2)
            return fail(RT_ERR_LIMIT, "scene BVH deeper than the kernel's traversal stack (" + std::to_string(bvh_depth) + "/" +
                                          std::to_string(P.light_bvh_depth) + ")");
        V.light_nodes = keep(upload(P.light_nodes, bytes));
        V.light_sep = keep(upload(P.light_sep, bytes));
        V.ref_nodes = keep(upload(P.ref_nodes, bytes));
        V.ref_light_nodes = keep(upload(P.ref_light_nodes, bytes));
        V.box_c2 = P.box_c2; V.box_c2x = 1.25f * P.box_c2;
        if (const char *e = getenv("RTAMD_C2X_SCALE")) V.box_c2x *= (float)atof(e); // experiment: the walkers' absolute look-behind (the gate's `seen` follows)
        // how far behind the best hit the walkers still look, relative to t (rt_exact.h)
        V.cull_k = getenv("RTAMD_CULL_K") ? (float)atof(getenv("RTAMD_CULL_K")) : 0.0078125f;
        V.exact_boxes = (getenv("RTAMD_NO_EXACT_BOXES") || fast_build) ? 0u : 1u; // RT_BUILD_DEVICE_BVH: there is no reference tree to be exact about
        if (getenv("RTAMD_DIAG_LOOKBEHIND_ONLY")) V.exact_boxes = 2u; // diagnostic (timing only, pixels NOT exact): the walkers look behind as with the gate, every hit stands
        V.n_tripwire_groups = (V.exact_boxes != 1u || getenv("RTAMD_NO_TRIPWIRES")) ? 0u : P.n_tripwire_groups; // part of the exactness machinery
        V.tripwires = V.n_tripwire_groups ? keep(upload(P.tripwires, bytes)) : nullptr;
        V.lights = keep(upload(P.lights, bytes));
        uint32_t n_light_walk_nodes = 0;
        {   // the light walker's own tree (rt_types.h: light_walk_nodes / lights_walk)
            const uint32_t nl = (uint32_t)P.lights.size();
            std::vector<LightRec> tagged = P.lights; // pad = light index << 1 | last-of-leaf (of the REFERENCE topology for now)
            for (uint32_t i = 0; i < nl; i++) tagged[i].isect.pad = (i << 1) | (tagged[i].isect.pad ? 1u : 0u);
            bool own_tree = nl >= 64 && !getenv("RTAMD_HOST_LIGHT_BVH");
            if (own_tree) {
                LightRec *d_tagged = upload(tagged, bytes);
                uint64_t scratch = 0;
                float *d_lbox = upload(P.light_walk_box, scratch);
                DeviceTree lt;
                LightRec *d_walk = nullptr;
                try {
                    lt = build_tree_on_device(d_lbox, nl, P.box_pad, 16); // the hits of a walk share its 24-entry column with the node stack
                    HIP_CHECK(hipMalloc((void **)&d_walk, (size_t)nl * sizeof(LightRec)));
                    gather_records(d_tagged, d_walk, lt, nl, sizeof(LightRec), 11, true); // word 11 = TriIsect::pad
                    HIP_CHECK(hipDeviceSynchronize());
                } catch (...) {
                    (void)hipFree(d_lbox); (void)hipFree(d_tagged); if (d_walk) (void)hipFree(d_walk);
                    free_device_tree(lt);
                    throw;
                }
                (void)hipFree(d_lbox); (void)hipFree(d_tagged);
                (void)hipFree(lt.order); (void)hipFree(lt.last); lt.order = nullptr; lt.last = nullptr;
                V.light_walk_nodes = keep(lt.nodes); V.lights_walk = keep(d_walk);
                bytes += (uint64_t)lt.n_nodes * sizeof(GpuNode);
                s->light_walk_depth = lt.depth;
                n_light_walk_nodes = lt.n_nodes;
            } else {
                V.light_walk_nodes = V.light_nodes;
                n_light_walk_nodes = (uint32_t)P.light_nodes.size();
                V.lights_walk = keep(upload(tagged, bytes));
                s->light_walk_depth = P.light_bvh_depth;
            }
        }
        V.materials = keep(upload(P.materials, bytes));
        V.images = keep(upload(P.images, bytes));
        V.texels = keep(upload(P.texels, bytes));
        std::vector<float> lut(P.srgb_lut, P.srgb_lut + 256);
        V.srgb_lut = keep(upload(lut, bytes));
        V.n_tris = desc->n_triangles;
        V.n_nodes = n_nodes;
        V.n_lights = (uint32_t)P.lights.size();
        V.n_components = P.lights.empty() ? 2u : 3u; // scene.cpp:65-74
        V.n_lights_f = (float)V.n_lights; V.n_components_f = (float)V.n_components;
        V.last_level_emission_only = 1;
        for (uint32_t i = 0; i < desc->n_materials; i++) {
            const rt_material &m = desc->materials[i];
            if (!(m.metallic_factor >= 0 && m.metallic_factor <= 1 && m.base_color[0] >= 0 && m.base_color[1] >= 0 && m.base_color[2] >= 0))
                V.last_level_emission_only = 0;
        }
        if (getenv("RTAMD_NO_LAST_LEVEL_SHORTCUT")) V.last_level_emission_only = 0;
        V.env_image = P.env_image;
        for (int k = 0; k < 3; k++) {
            V.cam_pos[k] = desc->camera.position[k]; V.cam_right[k] = desc->camera.right[k];
            V.cam_up[k] = desc->camera.up[k]; V.cam_fwd[k] = desc->camera.forward[k];
            V.bg[k] = desc->bg_color[k];
        }
        s->fov_y = desc->camera.fov_y;
        V.tan_fov_y = (float)std::tan((double)(desc->camera.fov_y / 2)); // scene.cpp:180 (host libm, like the reference)
        {   // the walk nodes of the persistent pipeline: both trees four wide on one 16-bit grid that also holds the camera (rt_types.h GpuNode4Q)
            float glo[3], ghi[3];
            for (int k = 0; k < 3; k++) glo[k] = ghi[k] = V.cam_pos[k];
            join_root_box(V.nodes, glo, ghi);
            if (n_light_walk_nodes) join_root_box(V.light_walk_nodes, glo, ghi);
            V.grid = make_node_grid(glo, ghi);
            uint32_t n4 = 0, n4l = 0, d4 = 0, d4l = 0;
            V.nodes4 = keep(widen_nodes(V.nodes, n_nodes, V.grid, n4, d4));
            V.light_walk_nodes4 = keep(widen_nodes(V.light_walk_nodes, n_light_walk_nodes, V.grid, n4l, d4l));
            bytes += ((uint64_t)n4 + n4l) * sizeof(GpuNode4Q);
            s->wide_depth = d4; s->wide_light_depth = d4l;
        }
        HIP_CHECK(hipMalloc((void **)&s->d_work_counter, 64));
        s->allocations.push_back(s->d_work_counter);
        HIP_CHECK(hipMalloc((void **)&s->d_counters, 512));
        s->allocations.push_back(s->d_counters);
        HIP_CHECK(hipEventCreate(&s->ev_start));
        HIP_CHECK(hipEventCreate(&s->ev_stop));
        HIP_CHECK(hipDeviceSynchronize());
        double t2 = now_ms();
        s->light_order = P.light_order;
        s->info.n_triangles = desc->n_triangles;
        s->info.n_lights = V.n_lights;
        s->info.n_bvh_nodes = n_nodes;
        s->info.n_light_bvh_nodes = (uint32_t)P.light_nodes.size();
        s->info.bvh_depth = bvh_depth;
        s->info.light_bvh_depth = P.light_bvh_depth;
        s->info.device_bytes = bytes;
        s->info.prep_ms = t1 - t0;
        s->info.upload_ms = t2 - t1;
        *out = s.release();
        return RT_OK;
    } catch (const HipError &e) {
        return fail(RT_ERR_HIP, e.what());
    } catch (const std::exception &e) {
        return fail(RT_ERR_INVALID_ARG, e.what());
    }
}

extern "C" {

void rt_scene_destroy(rt_scene *scene) { delete scene; }

int rt_scene_get_info(const rt_scene *scene, rt_scene_info *info) {
    if (!scene || !info) return fail(RT_ERR_INVALID_ARG, "rt_scene_get_info: null argument");
    *info = scene->info;
    return RT_OK;
}

int rt_scene_get_light_order(const rt_scene *scene, uint32_t *out, uint32_t capacity) {
    if (!scene || (!out && capacity)) return fail(RT_ERR_INVALID_ARG, "rt_scene_get_light_order: null argument");
    if (capacity < scene->light_order.size()) return fail(RT_ERR_INVALID_ARG, "rt_scene_get_light_order: buffer too small");
    memcpy(out, scene->light_order.data(), scene->light_order.size() * sizeof(uint32_t));
    return (int)scene->light_order.size();
}

int rt_host_prepare_orders(const rt_scene_desc *desc, int integrator, uint32_t *figure_order, uint32_t figure_capacity,
                           uint32_t *light_order, uint32_t light_capacity) {
    if (!desc || desc->struct_size != sizeof(rt_scene_desc)) return fail(RT_ERR_INVALID_ARG, "rt_host_prepare_orders: bad desc");
    try {
        std::vector<uint32_t> fo, lo;
        if (integrator == RT_INTEGRATOR_HW8 || integrator == RT_INTEGRATOR_HW7) { PreparedScene P; prepare_scene(*desc, P); fo = P.figure_order; lo = P.light_order; }
        else if (integrator == RT_INTEGRATOR_HW6) { PreparedScene6 P; prepare_scene_hw6(*desc, P); fo = P.figure_order; lo = P.light_order; }
        else if (integrator == RT_INTEGRATOR_HW5) { PreparedScene5 P; prepare_scene_hw5(*desc, P); fo = P.figure_order; lo = P.light_order; }
        else return fail(RT_ERR_UNSUPPORTED, "rt_host_prepare_orders: integrator must be HW5, HW6, HW7 or HW8");
        if ((figure_order && figure_capacity < fo.size()) || (light_order && light_capacity < lo.size())) return fail(RT_ERR_INVALID_ARG, "rt_host_prepare_orders: buffer too small");
        if (figure_order) memcpy(figure_order, fo.data(), fo.size() * sizeof(uint32_t));
        if (light_order) memcpy(light_order, lo.data(), lo.size() * sizeof(uint32_t));
        return (int)lo.size();
    } catch (const std::exception &e) {
        return fail(RT_ERR_INVALID_ARG, std::string("rt_host_prepare_orders: ") + e.what());
    }
}

static bool resolve_tiles(const rt_render_params *p, RenderView &R, std::string &err) {
    if (p->width <= 0 || p->height <= 0 || p->samples <= 0) { err = "width, height and samples must be positive"; return false; }
    if ((int64_t)p->width * p->height >= 2147483647LL) { err = "image too large for the per-pixel seed (y*W+x must stay below 2^31-1)"; return false; }
    R.width = p->width; R.height = p->height; R.samples = p->samples;
    R.ray_depth = p->ray_depth > 0 ? p->ray_depth : 6;
    if (R.ray_depth > RT_MAX_DEPTH) { err = "ray_depth above RT_MAX_DEPTH (16)"; return false; }
    R.shard_count = p->shard_count > 1 ? p->shard_count : 1;
    R.shard_index = p->shard_count > 1 ? p->shard_index : 0;
    if (R.shard_index < 0 || R.shard_index >= R.shard_count) { err = "shard_index out of range"; return false; }
    if (R.shard_count > 1) {
        R.tile_w = p->tile_w > 0 ? p->tile_w : 32;
        R.tile_h = p->tile_h > 0 ? p->tile_h : 32;
        if ((R.tile_w & 7) || (R.tile_h & 7)) { err = "tile_w and tile_h must be multiples of 8"; return false; }
    } else {
        R.tile_w = R.tile_h = 8;
    }
    R.tiles_x = (R.width + R.tile_w - 1) / R.tile_w;
    R.tiles_y = (R.height + R.tile_h - 1) / R.tile_h;
    uint32_t total = (uint32_t)R.tiles_x * (uint32_t)R.tiles_y;
    R.n_shard_tiles = total > (uint32_t)R.shard_index ? (total - (uint32_t)R.shard_index + (uint32_t)R.shard_count - 1) / (uint32_t)R.shard_count : 0;
    return true;
}

size_t rt_output_elems(const rt_render_params *p) {
    if (!p) return 0;
    RenderView R{};
    std::string err;
    if (!resolve_tiles(p, R, err)) return 0;
    if (R.shard_count > 1) return (size_t)R.n_shard_tiles * R.tile_w * R.tile_h * 3;
    return (size_t)R.width * R.height * 3;
}

// Rounds a pixel needs per camera sample: one per bounce; without the deepest-level shortcut the last bounce's pdf / clamp
// step takes one more.
static size_t wavefront_rounds(const SceneView &V, const RenderView &R) {
    return (size_t)R.samples * ((size_t)R.ray_depth + (V.last_level_emission_only ? 0u : 1u));
}

// Wavefront driver: per round one traverse launch and one shade launch (device/rt_wavefront.h).
// No host synchronisation inside: queue lengths live in device memory, one counter block per round.
// RTAMD_WF_PIPELINES=n (experiment, default 1): cut the frame into n independent pipelines (disjoint path slots, own queues and
// counters), each a chain traverse -> shade -> traverse ... on its own stream, so that one pipeline's kernels could fill the CUs
// another leaves idle in the tail of a launch.  Pixels do not depend on the cut (paths never interact; tests force n = 2..4).
// Measured on the benchmark frame: 2 pipelines 281, 3 pipelines 258 against 304 Msamples/s with one -- the halved queues lose
// more to ramp-up and drain than the overlap returns -- so one pipeline stays the default.
#define WF_MAX_PIPES 4
static int wavefront_pipelines(uint32_t n_work) {
    int p = 1;
    if (const char *e = getenv("RTAMD_WF_PIPELINES")) { int v = atoi(e); if (v >= 1 && v <= WF_MAX_PIPES) p = v; }
    while (p > 1 && n_work < (uint32_t)p) p--;
    return p;
}

static void launch_wavefront(rt_scene *scene, const SceneView &V, const RenderView &R, uint32_t n_work, hipStream_t stream, bool count, bool time_trace) {
    const size_t n_slots = (size_t)n_work * 64;
    const size_t rounds = wavefront_rounds(V, R);
    const int pipes = wavefront_pipelines(n_work);
    const size_t ctr_block = (rounds + 2) * WF_CTR; // words per pipeline
    if (scene->wf_slots < n_slots || scene->wf_levels < (size_t)R.ray_depth || scene->wf_ctr_words < ctr_block * pipes) {
        scene->free_wf();
        auto alloc = [&](size_t bytes) { void *p = nullptr; HIP_CHECK(hipMalloc(&p, bytes)); scene->wf_allocs.push_back(p); return p; };
        scene->wf.r0 = (float4 *)alloc(n_slots * (16 * WF_REC_BASE + 32 * (size_t)R.ray_depth));
        scene->wf.q_trace[0] = (uint32_t *)alloc(n_slots * 4);
        scene->wf.q_trace[1] = (uint32_t *)alloc(n_slots * 4);
        scene->wf.q_light = (uint32_t *)alloc(n_slots * 4);
        scene->wf.ctr = (uint32_t *)alloc(ctr_block * WF_MAX_PIPES * 4);
        scene->wf.q_slow = (uint32_t *)alloc(n_slots * 4);
        scene->wf_slots = n_slots; scene->wf_levels = (size_t)R.ray_depth; scene->wf_ctr_words = ctr_block * WF_MAX_PIPES;
    }
    scene->wf_pipes = pipes; scene->wf_ctr_block = ctr_block;
    // Trees deeper than the LDS stacks use the SPILL kernel variant (bounds-checked stack with a global overflow area).
    // RTAMD_WF_LDS_STACK=n (testing): pretend the LDS stacks hold only n entries, which forces the SPILL variant and its overflow area.
    int lds_limit = WF_STACK;
    if (const char *e = getenv("RTAMD_WF_LDS_STACK")) { int v = atoi(e); if (v >= 1 && v < WF_STACK) lds_limit = v; }
    const bool spill = scene->info.bvh_depth > (uint32_t)lds_limit || scene->info.light_bvh_depth > (uint32_t)lds_limit;
    const size_t ovf_block = (size_t)scene->n_cus * 8u * 256u * WF_OVF; // words per pipeline: up to 8 persistent blocks per CU
    if (spill && scene->wf_ovf_words < ovf_block * pipes) {
        void *p = nullptr;
        HIP_CHECK(hipMalloc(&p, ovf_block * pipes * 4u));
        scene->wf_allocs.push_back(p); // a smaller earlier area stays allocated until free_wf()
        scene->wf.ovf = (uint32_t *)p;
        scene->wf_ovf_words = ovf_block * pipes;
    }
    HIP_CHECK(hipMemsetAsync(scene->wf.ctr, 0, ctr_block * pipes * 4, stream));
    uint32_t blocks_per_cu = 5u;                                      // 5 x 30 KB of stacks fit the 160 KB LDS (which is handed out in 1280-byte granules: 32 KB blocks fit only 4 times)
    if (const char *e = getenv("RTAMD_WF_BLOCKS_PER_CU")) blocks_per_cu = (uint32_t)atoi(e) > 0 ? (uint32_t)atoi(e) : blocks_per_cu;
    if (blocks_per_cu > 8u) blocks_per_cu = 8u;
    const uint32_t persistent_blocks = (uint32_t)scene->n_cus * blocks_per_cu;
    int dyn256 = 64;                                                   // share of each queue (of 256) handed out dynamically at the tail
    if (const char *e = getenv("RTAMD_WF_DYNAMIC_256")) dyn256 = atoi(e) < 0 ? 0 : (atoi(e) > 255 ? 255 : atoi(e));
    if (const char *e = getenv("RTAMD_WF_STEAL_CHUNK")) dyn256 |= ((atoi(e) > 0 ? atoi(e) : 64) & 255) << 8; // tuning: items per dynamic chunk (default 64)
    int split_a = 7, split_b = 8; // cost weights closest-hit : light query for the block split (two sweeps: 7:8 is ~1 % ahead of 1:1 and 8:7)
    if (const char *e = getenv("RTAMD_WF_SPLIT")) { int a = 0, b = 0; if (sscanf(e, "%d:%d", &a, &b) == 2 && a > 0 && b > 0 && a < 256 && b < 256) { split_a = a; split_b = b; } }
    dyn256 |= (split_a << 16) | (split_b << 24);
    auto env_int = [](const char *n, int dflt) { const char *e = getenv(n); return e && atoi(e) > 0 ? atoi(e) : dflt; };
    // leaf batch: lanes waiting at a leaf start their triangle tests when 20 of them wait -- or, in a thinly populated wave, a share
    // of the active lanes (RTAMD_WF_LEAF_SHARE_256, default 112/256; sweep: tools/tuning/sweep_leaf_share.sh); packed as batch | share << 16
    const int leaf_share = env_int("RTAMD_WF_LEAF_SHARE_256", 112) & 0x7fff;
    const int t_refill = env_int("RTAMD_TRACE_REFILL", WF_REFILL), t_batch = (env_int("RTAMD_TRACE_LEAF_BATCH", WF_LEAF_BATCH) & 255) | (leaf_share << 16);
    const int l_refill = env_int("RTAMD_LIGHT_REFILL", WF_REFILL), l_batch = (env_int("RTAMD_LIGHT_LEAF_BATCH", WF_LEAF_BATCH) & 255) | (leaf_share << 16);
    const uint32_t shade_per_cu = (uint32_t)env_int("RTAMD_WF_SHADE_BLOCKS_PER_CU", 16); // grid cap of wf_shade_kernel (grid-stride beyond it)
    unsigned long long *ctrs = count ? scene->d_counters : nullptr;
    if (time_trace) while (scene->ev_pool.size() < 2 * rounds * pipes) { hipEvent_t e; HIP_CHECK(hipEventCreate(&e)); scene->ev_pool.push_back(e); }

    // the pipelines: contiguous ranges of 64-slot groups, the state arrays offset accordingly
    dev::WfView Wp[WF_MAX_PIPES];
    hipStream_t sp[WF_MAX_PIPES];
    uint32_t shade_blocks[WF_MAX_PIPES];
    if (pipes > 1 && !scene->wf_streams[0]) {
        for (int h = 0; h < WF_MAX_PIPES; h++) {
            HIP_CHECK(hipStreamCreateWithFlags(&scene->wf_streams[h], hipStreamNonBlocking));
            HIP_CHECK(hipEventCreateWithFlags(&scene->ev_join[h], hipEventDisableTiming));
        }
        HIP_CHECK(hipEventCreateWithFlags(&scene->ev_fork, hipEventDisableTiming));
    }
    const uint32_t stride = (uint32_t)WF_REC_BASE + 2u * (uint32_t)scene->wf_levels; // float4 per slot (the allocation's depth, >= this render's)
    uint32_t first = 0;
    for (int h = 0; h < pipes; h++) {
        const uint32_t groups = n_work / (uint32_t)pipes + ((uint32_t)h < n_work % (uint32_t)pipes ? 1u : 0u);
        const size_t base = (size_t)first * 64;
        dev::WfView W = scene->wf;
        W.stride = stride;
        W.r0 += base * stride;
        W.q_trace[0] += base; W.q_trace[1] += base; W.q_light += base; W.q_slow += base;
        W.ctr += (size_t)h * ctr_block;
        if (W.ovf) W.ovf += (size_t)h * ovf_block;
        W.n_slots = groups * 64u;
        W.slot_base = (uint32_t)base;
        Wp[h] = W;
        sp[h] = pipes > 1 ? scene->wf_streams[h] : stream;
        shade_blocks[h] = (W.n_slots + 255u) / 256u;
        if (shade_blocks[h] > (uint32_t)scene->n_cus * shade_per_cu) shade_blocks[h] = (uint32_t)scene->n_cus * shade_per_cu;
        first += groups;
    }
    if (pipes > 1) {
        HIP_CHECK(hipEventRecord(scene->ev_fork, stream));
        for (int h = 0; h < pipes; h++) HIP_CHECK(hipStreamWaitEvent(sp[h], scene->ev_fork, 0));
    }
    for (int h = 0; h < pipes; h++)
        hipLaunchKernelGGL(dev::wf_init_kernel, dim3((Wp[h].n_slots + 255u) / 256u), dim3(256), 0, sp[h], V, R, Wp[h]);
    const dim3 pb(persistent_blocks), tb(256);
    for (uint32_t r = 0; r < (uint32_t)rounds; r++) {
        for (int h = 0; h < pipes; h++) {
            const dev::WfView &W = Wp[h];
            hipStream_t st = sp[h];
            if (time_trace) HIP_CHECK(hipEventRecord(scene->ev_pool[2 * ((size_t)r * pipes + h)], st));
            if (spill) {
                if (count) hipLaunchKernelGGL((dev::wf_traverse_kernel<true, true>), pb, tb, 0, st, V, W, r, ctrs, t_refill, t_batch, l_refill, l_batch, dyn256, lds_limit);
                else hipLaunchKernelGGL((dev::wf_traverse_kernel<false, true>), pb, tb, 0, st, V, W, r, ctrs, t_refill, t_batch, l_refill, l_batch, dyn256, lds_limit);
            } else {
                if (count) hipLaunchKernelGGL((dev::wf_traverse_kernel<true, false>), pb, tb, 0, st, V, W, r, ctrs, t_refill, t_batch, l_refill, l_batch, dyn256, lds_limit);
                else hipLaunchKernelGGL((dev::wf_traverse_kernel<false, false>), pb, tb, 0, st, V, W, r, ctrs, t_refill, t_batch, l_refill, l_batch, dyn256, lds_limit);
            }
            if (time_trace) HIP_CHECK(hipEventRecord(scene->ev_pool[2 * ((size_t)r * pipes + h) + 1], st));
            if (!spill && V.n_lights) hipLaunchKernelGGL(dev::wf_light_exact_kernel, dim3((unsigned)scene->n_cus), dim3(64), 0, st, V, W, r);
            hipLaunchKernelGGL(dev::wf_shade_kernel, dim3(shade_blocks[h]), dim3(256), 0, st, V, R, W, r, ctrs);
            if (V.exact_boxes) hipLaunchKernelGGL(dev::wf_trace_exact_kernel, dim3((unsigned)scene->n_cus), dim3(64), 0, st, V, R, W, r, ctrs); // hits at a box boundary (~1e-5 of the paths)
        }
    }
    HIP_CHECK(hipGetLastError());
    if (pipes > 1)
        for (int h = 0; h < pipes; h++) {
            HIP_CHECK(hipEventRecord(scene->ev_join[h], sp[h]));
            HIP_CHECK(hipStreamWaitEvent(stream, scene->ev_join[h], 0));
        }
}

// Sample counts at which the launches of a persistent render stop (the last one = the frame).  Two phases: 1/16 of the samples under the
// round-robin deal, the rest after the re-deal.  RTAMD_PT_PHASES=3 deals the last quarter of the remaining samples once more, from the
// costs measured over the long middle phase — measured: no gain (hw6 practice6_2 191.7 vs 193.6 Msamples/s, headline 284.2 vs 284.6):
// what remains of the spread of the workgroups' exit times after one re-deal is not the amount of work but the serial samples of the
// slowest pixels.
static std::vector<int> phase_stops(bool rebalance, int phase0, int samples, int default_phases, bool small_population) {
    std::vector<int> stops;
    if (rebalance && getenv("RTAMD_PT_STOPS")) { // experiment: explicit sample counts at which the frame is re-dealt, e.g. "2,16"
        int last = 0;
        for (const char *c = getenv("RTAMD_PT_STOPS"); *c;) {
            const int v = atoi(c);
            if (v > last && v < samples) { stops.push_back(v); last = v; }
            while (*c && *c != ',') c++;
            if (*c == ',') c++;
        }
        stops.push_back(samples);
        return stops;
    }
    if (rebalance) {
        // small populations (the deal is in quarter sub-tiles): a first re-deal after two samples already — the round-robin deal of the
        // first phase is the costly one there (slowest workgroup / mean 1.8 on hw6's 1024x1024 frame) — then the usual one
        if (small_population && phase0 > 2 && !getenv("RTAMD_PT_PHASE0")) stops.push_back(2);
        stops.push_back(phase0);
        const int want = getenv("RTAMD_PT_PHASES") ? atoi(getenv("RTAMD_PT_PHASES")) : default_phases;
        const int mid = phase0 + (samples - phase0) * 3 / 4;
        if (want >= 3 && samples >= 64 && mid > phase0 && mid < samples) stops.push_back(mid);
    }
    stops.push_back(samples);
    return stops;
}

// Between the phases of a persistent render: read what every 8x8 sub-tile cost in the phase that just ended (PT_COST_*, rt_persistent.h)
// and how long every workgroup ran, and deal the sub-tiles again — longest processing time first onto the workgroup that would be done
// with it soonest (at most groups_per_block each); the lists go to d_ofs / d_ids for the next launch.
// Workgroups are not equally fast.  The five that share a CU are served oldest wave first, so at equal load they leave staggered
// (measured on the 1080p frame: 1,278 / 1,328 / 1,384 / 1,447 / 1,526 ms by launch order), and from the first exit on the CU runs with four
// waves per SIMD, then three ...  So the deal is by speed: speed(b) = (cost of the sub-tiles b owned) / (its run time) in the phase that
// just ended, and a sub-tile goes to the workgroup with the smallest load / speed.  `owner` (sub-tile -> workgroup of the phase that
// just ended; empty = the kernel's round-robin deal) is updated to the new deal.  d_times: per workgroup {start, exit, -} in 100 MHz
// ticks (PtParams::debug), nullable.
static void redeal_groups(rt_scene *scene, const uint32_t *d_cost, uint32_t *d_ofs, uint32_t *d_ids, uint32_t groups, uint32_t blocks, uint32_t groups_per_block, hipStream_t stream,
                          const unsigned long long *d_times = nullptr, std::vector<uint32_t> *owner = nullptr) {
    std::vector<uint32_t> cost(groups), ofs, ids, order(groups);
    HIP_CHECK(hipStreamSynchronize(stream));
    const double t0 = now_ms();
    HIP_CHECK(hipMemcpy(cost.data(), d_cost, (size_t)groups * 4, hipMemcpyDeviceToHost));
    std::vector<double> slowness(blocks, 1.0); // time per unit of cost, relative to the mean
    if (d_times && owner && !getenv("RTAMD_PT_NO_SPEEDS")) {
        std::vector<unsigned long long> times((size_t)blocks * 3);
        HIP_CHECK(hipMemcpy(times.data(), d_times, times.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        std::vector<double> had(blocks, 0.0);
        for (uint32_t g = 0; g < groups; g++) had[owner->empty() ? g % blocks : (*owner)[g]] += (double)cost[g] + 1.0;
        double sum = 0; uint32_t n = 0;
        for (uint32_t b = 0; b < blocks; b++) {
            const double dur = times[3 * b + 1] > times[3 * b] ? (double)(times[3 * b + 1] - times[3 * b]) : 0.0;
            slowness[b] = dur > 0 && had[b] > 0 ? dur / had[b] : 0.0;
            if (slowness[b] > 0) { sum += slowness[b]; n++; }
        }
        const double mean = n ? sum / n : 1.0;
        // The workgroups are dispatched in index order, one round of n_cus after the other, so the age rank of a workgroup on its CU is
        // its index / n_cus.  The measured slowness is split into the mean of the workgroup's round (the age effect: over-relaxed,
        // because a slow workgroup ran its last stretch with its faster neighbours already gone, so the phase shows less of a
        // difference than an even finish will: 2.0 measured best for the hw8 kernel on the 1080p frame — its youngest workgroups
        // still left last at 1.5 (exit times by dispatch round 1,028 / 1,024 / 1,035 / 1,071 / 1,118 ms) — and 1.5 for the hw6 kernel) and the
        // workgroup's own deviation from it (most of which is gone in the next phase: damped).
        const bool hw6_kernel = scene->flavor == RT_INTEGRATOR_HW6;
        const double gamma_round = getenv("RTAMD_PT_SPEED_GAMMA") ? atof(getenv("RTAMD_PT_SPEED_GAMMA")) : (hw6_kernel ? 1.5 : 2.0);
        const double gamma_own = getenv("RTAMD_PT_SPEED_GAMMA_OWN") ? atof(getenv("RTAMD_PT_SPEED_GAMMA_OWN")) : (hw6_kernel ? 0.4 : 0.3);
        const uint32_t round_size = scene->n_cus > 0 && blocks % (uint32_t)scene->n_cus == 0 ? (uint32_t)scene->n_cus : blocks;
        for (uint32_t r0 = 0; r0 < blocks; r0 += round_size) {
            double lsum = 0; uint32_t ln = 0;
            for (uint32_t b = r0; b < r0 + round_size; b++) if (slowness[b] > 0) { lsum += log(slowness[b] / mean); ln++; }
            const double lround = ln ? lsum / ln : 0.0;
            for (uint32_t b = r0; b < r0 + round_size; b++) {
                const double s = slowness[b] > 0 ? exp(gamma_round * lround + gamma_own * (log(slowness[b] / mean) - lround)) : exp(gamma_round * lround);
                slowness[b] = s < 0.5 ? 0.5 : (s > 2.0 ? 2.0 : s); // a workgroup with almost nothing to do says little about its speed
            }
        }
    }
    for (uint32_t g = 0; g < groups; g++) order[g] = g;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return cost[a] != cost[b] ? cost[a] > cost[b] : a < b; });
    std::vector<std::pair<double, uint32_t>> heap; // (time at which the block is done with what it holds, block), min-heap
    heap.reserve(blocks);
    for (uint32_t b = 0; b < blocks; b++) heap.push_back({0.0, b});
    auto cmp = [](const std::pair<double, uint32_t> &a, const std::pair<double, uint32_t> &b) { return a > b; };
    std::make_heap(heap.begin(), heap.end(), cmp);
    std::vector<std::vector<uint32_t>> mine(blocks);
    uint64_t total = 0, before_max = 0;
    { std::vector<uint64_t> rr(blocks, 0); for (uint32_t g = 0; g < groups; g++) { rr[g % blocks] += cost[g]; total += cost[g]; } for (uint64_t v : rr) before_max = v > before_max ? v : before_max; }
    for (uint32_t g : order) {
        std::pop_heap(heap.begin(), heap.end(), cmp);
        auto &top = heap.back();
        mine[top.second].push_back(g);
        top.first += ((double)cost[g] + 1.0) * slowness[top.second]; // + 1: empty sub-tiles (padding) are spread evenly too
        if (mine[top.second].size() >= groups_per_block) heap.pop_back(); // full: out of the deal
        else std::push_heap(heap.begin(), heap.end(), cmp);
    }
    ofs.assign(blocks + 1, 0);
    if (owner) owner->assign(groups, 0);
    const bool by_cost = !getenv("RTAMD_PT_NO_FRONT_FIRST"); // most expensive group first: the kernel's queues serve the front of the list first (PtParams::front_first)
    for (uint32_t b = 0; b < blocks; b++) {
        if (by_cost) std::sort(mine[b].begin(), mine[b].end(), [&](uint32_t x, uint32_t y) { return cost[x] != cost[y] ? cost[x] > cost[y] : x < y; });
        else std::sort(mine[b].begin(), mine[b].end());
        ofs[b] = (uint32_t)ids.size();
        ids.insert(ids.end(), mine[b].begin(), mine[b].end());
        if (owner) for (uint32_t g : mine[b]) (*owner)[g] = b;
    }
    ofs[blocks] = (uint32_t)ids.size();
    if (const char *dump = getenv("RTAMD_DUMP_DEAL")) { // diagnostic: what the re-deal gave every workgroup (tools/tuning/wg_balance.py)
        if (FILE *f = fopen(dump, "w")) {
            for (uint32_t b = 0; b < blocks; b++) {
                uint64_t load = 0;
                for (uint32_t g : mine[b]) load += cost[g];
                fprintf(f, "%u %zu %llu %.4f\n", b, mine[b].size(), (unsigned long long)load, slowness[b]);
            }
            fclose(f);
        }
    }
    HIP_CHECK(hipMemcpyAsync(d_ofs, ofs.data(), ofs.size() * 4, hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(d_ids, ids.data(), ids.size() * 4, hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipStreamSynchronize(stream)); // the vectors go out of scope
    scene->pt_rebalance_ms += now_ms() - t0;
    if (total) scene->pt_imbalance = (double)before_max * blocks / (double)total; // slowest workgroup / mean under the round-robin deal
}

// Persistent dataflow driver (device/rt_persistent.h): ONE launch renders up to n_cus x PT_MAX_PATHS path slots; larger frames
// (or throughput mode with many streams) take several passes over disjoint slot ranges, each a complete render of its pixels.
//
// Load balance.  A workgroup owns its pixels for a whole launch, and pixels differ in cost (sky: one query per sample, a glossy
// interior: up to 2 x depth), so with the plain round-robin deal of the 8x8 sub-tiles the slowest workgroup ends 3 % (1080p on
// one GPU) to 12 % (a shard of 8) behind the mean.  A frame of >= 16 samples is therefore rendered in two phases: the first
// 1/16 of the samples with the round-robin deal while every workgroup counts the hits it shades per sub-tile, then the host
// re-deals the sub-tiles (longest processing time first onto the least loaded workgroup) and the second launch resumes every
// pixel from its record (pixel sum, random stream and the parked camera ray are all there; pixels do not depend on the deal).
// Every launch is bracketed by events when `time_trace` (ev_pool[2k], ev_pool[2k+1]).
static void launch_persistent(rt_scene *scene, const SceneView &V, const RenderView &R, uint32_t n_work, hipStream_t stream, bool count, bool time_trace) {
    auto env_int = [](const char *n, int dflt) { const char *e = getenv(n); return e && atoi(e) > 0 ? atoi(e) : dflt; };
    uint32_t n_blocks_max = (uint32_t)env_int("RTAMD_PT_BLOCKS", scene->n_cus * P8_PER_CU); // five 4-wave workgroups per CU (their LDS fills the CU)
    const uint64_t pass_cap = (uint64_t)n_blocks_max * (PT_MAX_PATHS / 64);
    const uint32_t passes = (uint32_t)((n_work + pass_cap - 1) / pass_cap);
    const uint32_t pass_groups = (n_work + passes - 1) / passes;         // 8x8 sub-tiles (64 path slots) per pass
    // The unit of the deal: an 8x8 sub-tile, or — when a workgroup would hold fewer than sixteen of those (small frames, shards) — a
    // quarter of one (two pixel rows): a single heavy sub-tile must not outweigh a workgroup's fair share, and a workgroup whose load
    // is a few heavy pixels is bound by their serial samples.
    uint32_t group_shift = (uint64_t)pass_groups < 16ull * n_blocks_max ? 4u : 6u;
    if (const char *e = getenv("RTAMD_PT_GROUP_SHIFT")) { const int v = atoi(e); if (v == 4 || v == 5 || v == 6) group_shift = (uint32_t)v; }
    const uint32_t sub = 6u - group_shift, groups_per_block = PT_MAX_PATHS >> group_shift;
    const size_t n_slots = (size_t)pass_groups * 64;
    if (scene->pt_slots < n_slots || scene->pt_levels < (size_t)R.ray_depth) {
        if (scene->pt_r0) (void)hipFree(scene->pt_r0);
        scene->pt_r0 = nullptr; scene->pt_slots = scene->pt_levels = 0;
        HIP_CHECK(hipMalloc((void **)&scene->pt_r0, n_slots * (16 * WF_REC_BASE + 32 * (size_t)R.ray_depth)));
        scene->pt_slots = n_slots; scene->pt_levels = (size_t)R.ray_depth;
    }
    const size_t group_words = 2 * ((size_t)pass_groups << sub) + n_blocks_max + 1;
    if (scene->pt_group_words < group_words) {
        if (scene->pt_groups) (void)hipFree(scene->pt_groups);
        scene->pt_groups = nullptr; scene->pt_group_words = 0;
        HIP_CHECK(hipMalloc((void **)&scene->pt_groups, group_words * 4));
        scene->pt_group_words = group_words;
    }
    uint32_t *d_cost = scene->pt_groups, *d_ofs = d_cost + ((size_t)pass_groups << sub), *d_ids = d_ofs + n_blocks_max + 1;
    dev::PtParams P{};
    const int leaf_share = env_int("RTAMD_WF_LEAF_SHARE_256", 112) & 0x7fff;
    P.refill = env_int("RTAMD_TRACE_REFILL", WF_REFILL) | (env_int("RTAMD_LIGHT_REFILL", env_int("RTAMD_TRACE_REFILL", WF_REFILL)) << 16);
    P.leaf_batch = (env_int("RTAMD_TRACE_LEAF_BATCH", 28) & 255) | (leaf_share << 16); // lanes that hold two leaves (or have nothing else left) before a leaf phase starts
    P.shade_min = 0; // set per pass below
    P.shade_thr0 = env_int("RTAMD_PT_SHADE_THR0", 128);
    P.shade_thr_step = env_int("RTAMD_PT_SHADE_STEP", 512);
    P.cost_t = 7; P.cost_l = 8;
    if (const char *e = getenv("RTAMD_WF_SPLIT")) { int a = 0, b = 0; if (sscanf(e, "%d:%d", &a, &b) == 2 && a > 0 && b > 0 && a < 256 && b < 256) { P.cost_t = a; P.cost_l = b; } }
    P.prio = getenv("RTAMD_PT_PRIO") ? atoi(getenv("RTAMD_PT_PRIO")) : 0;
    P.counters = scene->d_counters;
    // A wave still in the launch after this long gives up (the kernel cannot hang the GPU): RTAMD_PT_TIMEOUT_S, by default ten minutes or
    // — for long renders: 4K at thousands of samples — the time the launch would take at a twentieth of the usual rate, whichever is more.
    {
        const double expected_s = (double)n_work * 64.0 * (double)R.samples / 15e6;
        const double deadline_s = getenv("RTAMD_PT_TIMEOUT_S") ? (double)env_int("RTAMD_PT_TIMEOUT_S", 600) : (expected_s > 600.0 ? expected_s : 600.0);
        P.deadline_ticks = (unsigned long long)(deadline_s * 1e8);
    }
    // every workgroup leaves its start and exit time (the re-deal measures the workgroups' speeds with them)
    P.debug = nullptr;
    if (!scene->d_pt_debug) HIP_CHECK(hipMalloc((void **)&scene->d_pt_debug, (size_t)PT_DEBUG_BLOCKS * 3 * sizeof(unsigned long long)));
    if (n_blocks_max <= PT_DEBUG_BLOCKS) P.debug = scene->d_pt_debug;
    float4 *d_trace = nullptr;
    const uint32_t trace_cap = 1u << 16;
    if (count && getenv("RTAMD_TRACE_PIXEL") && getenv("RTAMD_TRACE_OUT")) { // diagnostic: tests/diagnostics/trace_pixel.py
        int tx = 0, ty = 0;
        if (sscanf(getenv("RTAMD_TRACE_PIXEL"), "%d,%d", &tx, &ty) == 2) {
            HIP_CHECK(hipMalloc((void **)&d_trace, (size_t)trace_cap * sizeof(float4)));
            HIP_CHECK(hipMemsetAsync(d_trace, 0, sizeof(float4), stream));
            P.trace_buf = d_trace; P.trace_cap = trace_cap; P.trace_pixel = ty * R.width + tx;
        }

    }
    // two phases when there is something to re-deal: enough samples, and several sub-tiles per workgroup
    const int phase0 = getenv("RTAMD_PT_PHASE0") ? atoi(getenv("RTAMD_PT_PHASE0")) : R.samples / 16;
    const bool two_phase = !getenv("RTAMD_PT_NO_REBALANCE") && phase0 >= 1 && phase0 < R.samples && ((uint64_t)pass_groups << sub) >= 4ull * n_blocks_max;
    const std::vector<int> stops = phase_stops(two_phase, phase0, R.samples, 2, group_shift < 6u);
    const uint32_t phases = (uint32_t)stops.size();
    if (time_trace) while (scene->ev_pool.size() < 2 * (size_t)passes * phases) { hipEvent_t e; HIP_CHECK(hipEventCreate(&e)); scene->ev_pool.push_back(e); }
    dev::WfView W{};
    W.r0 = scene->pt_r0;
    W.stride = (uint32_t)WF_REC_BASE + 2u * (uint32_t)scene->pt_levels;
    uint32_t first = 0, launch = 0;
    scene->pt_blocks = 0; scene->pt_rebalance_ms = 0; scene->pt_imbalance = 0;
    std::vector<uint32_t> owner; // sub-tile -> workgroup of the phase in flight
    for (uint32_t p = 0; p < passes; p++) {
        const uint32_t groups = n_work - first < pass_groups ? n_work - first : pass_groups;
        W.n_slots = groups * 64u;
        W.slot_base = first * 64u;
        const uint32_t n_units = groups << sub;   // groups of the deal
        P.n_groups = n_units; P.group_shift = group_shift;
        const uint32_t blocks = n_units < n_blocks_max ? n_units : n_blocks_max;
        if (blocks > scene->pt_blocks) scene->pt_blocks = blocks;
        // a wave turns shader when this many paths wait: a pool of a few hundred paths cannot let its paths wait for a full wave of them
        // (measured: 1,620 paths per workgroup 32 > 64 > 16; 820 and 200 paths per workgroup 16 > 32 > 64)
        P.shade_min = env_int("RTAMD_PT_SHADE_MIN", ((uint64_t)groups * 64u) / blocks >= 1536u ? 32 : 16);
        for (uint32_t ph = 0; ph < phases; ph++) {
            RenderView Rp = R;
            Rp.sample_stop = stops[ph];
            P.resume = ph ? 1u : 0u;
            P.group_cost = ph + 1 < phases ? d_cost : nullptr;   // every phase but the last measures for the next re-deal
            P.group_ofs = ph ? d_ofs : nullptr;
            P.group_ids = ph ? d_ids : nullptr;
            P.front_first = ph && !getenv("RTAMD_PT_NO_FRONT_FIRST") ? 1u : 0u;
            if (ph == 0) owner.clear(); // the kernel's round-robin deal
            if (ph >= 1) redeal_groups(scene, d_cost, d_ofs, d_ids, n_units, blocks, groups_per_block, stream, P.debug, &owner);
            if (P.debug) HIP_CHECK(hipMemsetAsync(scene->d_pt_debug, 0, (size_t)PT_DEBUG_BLOCKS * 3 * sizeof(unsigned long long), stream));
            if (time_trace) HIP_CHECK(hipEventRecord(scene->ev_pool[2 * launch], stream));
            // kernel variant by the features this render can reach (fewer features, fewer spilled registers): the hw7 integrator has no
            // environment map; an hw8 render needs the environment lookup only when the scene has a map
            const int feat = V.hw7 ? WF_FEAT_HW7 : (V.env_image >= 0 ? WF_FEAT_ENV : 0);
            const dim3 grid(blocks), block(P8_THREADS);
            if (count) {
                if (feat == WF_FEAT_HW7) hipLaunchKernelGGL((dev::pt_persistent_kernel<true, WF_FEAT_HW7>), grid, block, 0, stream, V, Rp, W, P);
                else if (feat == WF_FEAT_ENV) hipLaunchKernelGGL((dev::pt_persistent_kernel<true, WF_FEAT_ENV>), grid, block, 0, stream, V, Rp, W, P);
                else hipLaunchKernelGGL((dev::pt_persistent_kernel<true, 0>), grid, block, 0, stream, V, Rp, W, P);
            } else {
                if (feat == WF_FEAT_HW7) hipLaunchKernelGGL((dev::pt_persistent_kernel<false, WF_FEAT_HW7>), grid, block, 0, stream, V, Rp, W, P);
                else if (feat == WF_FEAT_ENV) hipLaunchKernelGGL((dev::pt_persistent_kernel<false, WF_FEAT_ENV>), grid, block, 0, stream, V, Rp, W, P);
                else hipLaunchKernelGGL((dev::pt_persistent_kernel<false, 0>), grid, block, 0, stream, V, Rp, W, P);
            }
            if (time_trace) HIP_CHECK(hipEventRecord(scene->ev_pool[2 * launch + 1], stream));
            launch++;
        }
        first += groups;
    }
    HIP_CHECK(hipGetLastError());
    scene->pt_passes = passes;
    scene->pt_launches = launch;
    if (d_trace) { // diagnostic dump: raw float32, 4 words header (count first), then 16 words per consumed hit record
        std::vector<float4> h(trace_cap);
        HIP_CHECK(hipStreamSynchronize(stream));
        HIP_CHECK(hipMemcpy(h.data(), d_trace, (size_t)trace_cap * sizeof(float4), hipMemcpyDeviceToHost));
        (void)hipFree(d_trace);
        if (FILE *f = fopen(getenv("RTAMD_TRACE_OUT"), "wb")) { fwrite(h.data(), sizeof(float4), trace_cap, f); fclose(f); }
    }
}

// hw6 in the persistent organisation (device/rt_persistent_hw6.h): passes of up to n_cus x P6_MAX_PATHS path slots, each rendered in
// two phases with a re-deal of the sub-tiles in between like launch_persistent — here the cost of a pixel spans 1 to 63 walks per
// sample (a wall against the glass bunny), so the deal matters far more than for hw8.
static void launch_persistent6(rt_scene *scene, const SceneView6 &V, const RenderView &R, uint32_t n_work, hipStream_t stream, bool count, bool time_trace) {
    auto env_int = [](const char *n, int dflt) { const char *e = getenv(n); return e && atoi(e) > 0 ? atoi(e) : dflt; };
    const uint32_t n_blocks_max = (uint32_t)env_int("RTAMD_PT_BLOCKS", scene->n_cus * P6_PER_CU); // five 4-wave workgroups per CU
    const uint64_t pass_cap = (uint64_t)n_blocks_max * (P6_MAX_PATHS / 64);
    const uint32_t passes = (uint32_t)((n_work + pass_cap - 1) / pass_cap);
    const uint32_t pass_groups = (n_work + passes - 1) / passes;         // 8x8 sub-tiles (64 path slots) per pass
    // The unit of the deal: an 8x8 sub-tile, or — when a workgroup would hold fewer than sixteen of those (small frames, shards) — a
    // quarter of one (two pixel rows): a single heavy sub-tile must not outweigh a workgroup's fair share, and a workgroup whose load
    // is a few heavy pixels is bound by their serial samples.
    uint32_t group_shift = (uint64_t)pass_groups < 16ull * n_blocks_max ? 4u : 6u;
    if (const char *e = getenv("RTAMD_PT_GROUP_SHIFT")) { const int v = atoi(e); if (v == 4 || v == 5 || v == 6) group_shift = (uint32_t)v; }
    const uint32_t sub = 6u - group_shift, groups_per_block = P6_MAX_PATHS >> group_shift;
    const size_t n_slots = (size_t)pass_groups * 64;
    if (scene->pt6_slots < n_slots) {
        if (scene->pt6_r0) (void)hipFree(scene->pt6_r0);
        scene->pt6_r0 = nullptr; scene->pt6_slots = 0;
        HIP_CHECK(hipMalloc((void **)&scene->pt6_r0, n_slots * (size_t)P6_REC * sizeof(float4)));
        scene->pt6_slots = n_slots;
    }
    const size_t group_words = 2 * ((size_t)pass_groups << sub) + n_blocks_max + 1;
    if (scene->pt_group_words < group_words) {
        if (scene->pt_groups) (void)hipFree(scene->pt_groups);
        scene->pt_groups = nullptr; scene->pt_group_words = 0;
        HIP_CHECK(hipMalloc((void **)&scene->pt_groups, group_words * 4));
        scene->pt_group_words = group_words;
    }
    uint32_t *d_cost = scene->pt_groups, *d_ofs = d_cost + ((size_t)pass_groups << sub), *d_ids = d_ofs + n_blocks_max + 1;
    dev::PtParams P{};
    const int leaf_share = env_int("RTAMD_WF_LEAF_SHARE_256", 112) & 0x7fff;
    P.refill = env_int("RTAMD_TRACE_REFILL", WF_REFILL) | (env_int("RTAMD_LIGHT_REFILL", env_int("RTAMD_TRACE_REFILL", WF_REFILL)) << 16);
    P.leaf_batch = (env_int("RTAMD_TRACE_LEAF_BATCH", 28) & 255) | (leaf_share << 16); // lanes that hold two leaves (or have nothing else left) before a leaf phase starts
    P.shade_min = 0; // set per pass below
    P.shade_thr0 = env_int("RTAMD_PT_SHADE_THR0", 128);
    P.shade_thr_step = env_int("RTAMD_PT_SHADE_STEP", 512);
    P.cost_t = 1; P.cost_l = 1;
    P.counters = scene->d_counters;
    // A wave still in the launch after this long gives up (the kernel cannot hang the GPU): RTAMD_PT_TIMEOUT_S, by default ten minutes or
    // — for long renders: 4K at thousands of samples — the time the launch would take at a twentieth of the usual rate, whichever is more.
    {
        const double expected_s = (double)n_work * 64.0 * (double)R.samples / 15e6;
        const double deadline_s = getenv("RTAMD_PT_TIMEOUT_S") ? (double)env_int("RTAMD_PT_TIMEOUT_S", 600) : (expected_s > 600.0 ? expected_s : 600.0);
        P.deadline_ticks = (unsigned long long)(deadline_s * 1e8);
    }
    // every workgroup leaves its start and exit time (the re-deal measures the workgroups' speeds with them)
    if (!scene->d_pt_debug) HIP_CHECK(hipMalloc((void **)&scene->d_pt_debug, (size_t)PT_DEBUG_BLOCKS * 3 * sizeof(unsigned long long)));
    if (n_blocks_max <= PT_DEBUG_BLOCKS) P.debug = scene->d_pt_debug;
    const int phase0 = getenv("RTAMD_PT_PHASE0") ? atoi(getenv("RTAMD_PT_PHASE0")) : R.samples / 16;
    const bool two_phase = !getenv("RTAMD_PT_NO_REBALANCE") && phase0 >= 1 && phase0 < R.samples && ((uint64_t)pass_groups << sub) >= 4ull * n_blocks_max;
    const std::vector<int> stops = phase_stops(two_phase, phase0, R.samples, 2, group_shift < 6u);
    const uint32_t phases = (uint32_t)stops.size();
    if (time_trace) while (scene->ev_pool.size() < 2 * (size_t)passes * phases) { hipEvent_t e; HIP_CHECK(hipEventCreate(&e)); scene->ev_pool.push_back(e); }
    dev::W6View W{};
    W.r0 = scene->pt6_r0;
    uint32_t first = 0, launch = 0;
    scene->pt_rebalance_ms = 0; scene->pt_imbalance = 0; scene->pt_blocks = 0;
    std::vector<uint32_t> owner; // sub-tile -> workgroup of the phase in flight
    for (uint32_t p = 0; p < passes; p++) {
        const uint32_t groups = n_work - first < pass_groups ? n_work - first : pass_groups;
        W.slot_base = first * 64u;
        const uint32_t n_units = groups << sub;   // groups of the deal
        P.n_groups = n_units; P.group_shift = group_shift;
        const uint32_t blocks = n_units < n_blocks_max ? n_units : n_blocks_max;
        if (blocks > scene->pt_blocks) scene->pt_blocks = blocks;
        // a wave turns shader when this many paths wait: a pool of a few hundred paths cannot let its paths wait for a full wave of them
        // (measured: 1,620 paths per workgroup 32 > 64 > 16; 820 and 200 paths per workgroup 16 > 32 > 64)
        P.shade_min = env_int("RTAMD_PT_SHADE_MIN", ((uint64_t)groups * 64u) / blocks >= 1536u ? 32 : 16);
        for (uint32_t ph = 0; ph < phases; ph++) {
            RenderView Rp = R;
            Rp.sample_stop = stops[ph];
            P.resume = ph ? 1u : 0u;
            P.group_cost = ph + 1 < phases ? d_cost : nullptr;   // every phase but the last measures for the next re-deal
            P.group_ofs = ph ? d_ofs : nullptr;
            P.group_ids = ph ? d_ids : nullptr;
            P.front_first = ph && !getenv("RTAMD_PT_NO_FRONT_FIRST") ? 1u : 0u;
            if (ph == 0) owner.clear(); // the kernel's round-robin deal
            if (ph >= 1) redeal_groups(scene, d_cost, d_ofs, d_ids, n_units, blocks, groups_per_block, stream, P.debug, &owner);
            if (P.debug) HIP_CHECK(hipMemsetAsync(scene->d_pt_debug, 0, (size_t)PT_DEBUG_BLOCKS * 3 * sizeof(unsigned long long), stream));
            if (time_trace) HIP_CHECK(hipEventRecord(scene->ev_pool[2 * launch], stream));
            if (count) hipLaunchKernelGGL(dev::p6_persistent_kernel<true>, dim3(blocks), dim3(P6_THREADS), 0, stream, V, Rp, W, P);
            else hipLaunchKernelGGL(dev::p6_persistent_kernel<false>, dim3(blocks), dim3(P6_THREADS), 0, stream, V, Rp, W, P);
            if (time_trace) HIP_CHECK(hipEventRecord(scene->ev_pool[2 * launch + 1], stream));
            launch++;
        }
        first += groups;
    }
    HIP_CHECK(hipGetLastError());
    scene->pt_passes = passes;
    scene->pt_launches = launch;
}

int rt_render(rt_scene *scene, const rt_render_params *p, float *out_rgb, uint8_t *out_rgb8, rt_stats *stats) {
    if (!scene || !p) return fail(RT_ERR_INVALID_ARG, "rt_render: null argument");
    if (p->struct_size != sizeof(rt_render_params)) return fail(RT_ERR_INVALID_ARG, "rt_render: struct_size mismatch (ABI skew)");
    if (p->integrator != RT_INTEGRATOR_HW8 && p->integrator != RT_INTEGRATOR_HW6 && p->integrator != RT_INTEGRATOR_HW3 && p->integrator != RT_INTEGRATOR_HW1 &&
        p->integrator != RT_INTEGRATOR_HW2 && p->integrator != RT_INTEGRATOR_HW4 && p->integrator != RT_INTEGRATOR_HW5 && p->integrator != RT_INTEGRATOR_HW7)
        return fail(RT_ERR_UNSUPPORTED, "rt_render: unknown integrator");
    const bool txt_scene = scene->flavor == RT_INTEGRATOR_HW3;
    const bool txt_integrator = p->integrator >= RT_INTEGRATOR_HW1 && p->integrator <= RT_INTEGRATOR_HW5;
    const bool hw7 = p->integrator == RT_INTEGRATOR_HW7; // renders a scene prepared for hw8 with hw7's material model (no textures)
    if (txt_scene != txt_integrator || (!txt_scene && p->integrator != scene->flavor && !(hw7 && scene->flavor == RT_INTEGRATOR_HW8)))
        return fail(RT_ERR_INVALID_ARG, "rt_render: this scene was prepared for integrator " + std::to_string(scene->flavor) +
                                            " (hw6 scenes carry no vertex normals, hw8 scenes do)");
    RenderView R{};
    std::string err;
    if (!resolve_tiles(p, R, err)) return fail(RT_ERR_INVALID_ARG, "rt_render: " + err);
    double t0 = now_ms();
    float *d_rgb = nullptr;
    uint8_t *d_rgb8 = nullptr;
    bool own_rgb = false, own_rgb8 = false;
    OwnedDev rgb_buf, rgb8_buf;
    try {
        HIP_CHECK(hipSetDevice(scene->device));
        hipStream_t stream = (hipStream_t)p->stream;
        const bool out_dev = (p->flags & RT_FLAG_OUT_DEVICE) != 0;
        const bool count = (p->flags & RT_FLAG_COUNTERS) != 0;
        size_t elems = rt_output_elems(p);
        if (out_rgb) {
            if (out_dev) d_rgb = out_rgb;
            else { HIP_CHECK(hipMalloc(&rgb_buf.p, elems * sizeof(float))); d_rgb = (float *)rgb_buf.p; own_rgb = true; }
        }
        if (out_rgb8) {
            if (out_dev) d_rgb8 = out_rgb8;
            else { HIP_CHECK(hipMalloc(&rgb8_buf.p, elems)); d_rgb8 = (uint8_t *)rgb8_buf.p; own_rgb8 = true; }
        }
        R.out_rgb = d_rgb; R.out_rgb8 = d_rgb8;
        R.work_counter = scene->d_work_counter;
        R.counters = count ? scene->d_counters : nullptr;
        // scene.cpp:181,176 — evaluated on the host in float exactly like the reference
        R.tan_fov_x = scene->view.tan_fov_y * R.width / R.height;
        R.inv_samples = (float)(1.0 / R.samples);
        R.sample_stop = R.samples;
        uint32_t n_work = R.n_shard_tiles * (uint32_t)((R.tile_w >> 3) * (R.tile_h >> 3));
        const int streams = p->sample_streams > 1 ? p->sample_streams : 1;
        if (p->reserved != 0) return fail(RT_ERR_INVALID_ARG, "rt_render: reserved must be 0");
        if ((p->flags & (RT_FLAG_SAMPLE_SEEDS | RT_FLAG_RUSSIAN_ROULETTE)) && streams <= 1)
            return fail(RT_ERR_INVALID_ARG, "rt_render: RT_FLAG_SAMPLE_SEEDS / RT_FLAG_RUSSIAN_ROULETTE change the estimator and belong to throughput mode (sample_streams > 1)");
        if (streams > 1) { // throughput mode (include/rtamd.h: sample_streams)
            if (p->integrator != RT_INTEGRATOR_HW8 && p->integrator != RT_INTEGRATOR_HW7 && p->integrator != RT_INTEGRATOR_HW6) return fail(RT_ERR_UNSUPPORTED, "rt_render: sample_streams > 1 is implemented for RT_INTEGRATOR_HW6 / HW7 / HW8 only");
            if (p->integrator == RT_INTEGRATOR_HW6 && (p->flags & (RT_FLAG_SAMPLE_SEEDS | RT_FLAG_RUSSIAN_ROULETTE))) return fail(RT_ERR_UNSUPPORTED, "rt_render: RT_FLAG_SAMPLE_SEEDS / RT_FLAG_RUSSIAN_ROULETTE are implemented for RT_INTEGRATOR_HW7 / HW8 only");
            if (streams > 256 || R.samples % streams != 0) return fail(RT_ERR_INVALID_ARG, "rt_render: samples must be a multiple of sample_streams (at most 256 streams)");
            if ((int64_t)R.width * R.height * streams >= 2147483647LL) return fail(RT_ERR_INVALID_ARG, "rt_render: width*height*sample_streams must stay below 2^31-1 (stream seeds)");
            if ((uint64_t)n_work * 64u * (uint64_t)streams >= 0x40000000ull) return fail(RT_ERR_LIMIT, "rt_render: too many path slots (pixels of this shard x sample_streams)");
        }
        HIP_CHECK(hipMemsetAsync(scene->d_work_counter, 0, 4, stream));
        HIP_CHECK(hipMemsetAsync(scene->d_counters, 0, 512, stream));
        if (count) {
            if (getenv("RTAMD_DEBUG_COUNTERS")) HIP_CHECK(hipMemsetAsync(scene->d_counters + 15, 1, 1, stream)); // asks the counting kernels for the in-flight histograms
        }
        uint32_t blocks = (uint32_t)scene->n_cus * 16u;
        if (blocks > n_work) blocks = n_work;
        // Kernel organisation: "wavefront" (default) or the single persistent "megakernel" (RTAMD_KERNEL=mega,
        // also the fallback when a BVH is deeper than the wavefront kernels' LDS stacks).
        const char *ksel = getenv("RTAMD_KERNEL");
        bool use_wavefront = !(ksel && strcmp(ksel, "mega") == 0);
        if (scene->info.bvh_depth > WF_STACK + WF_OVF || scene->info.light_bvh_depth > 64 || scene->info.n_triangles >= 0x40000000u) use_wavefront = false; // light depth: 64-bit frame mask
        if (scene->flavor == RT_INTEGRATOR_HW6 || txt_scene) use_wavefront = false;
        if (R.samples / streams >= (1 << 25)) use_wavefront = false; // the path record keeps the sample index in 25 bits
        // persistent dataflow pipeline (default) | round pipeline (RTAMD_KERNEL=wavefront, and for trees deeper than the LDS stack columns)
        // The persistent pipeline is the default at every size: it takes reference-exact box decisions at no measurable cost for the exact
        // walks themselves (they hide behind the other waves of the workgroup), keeps a small path population — a shard of a multi-GPU
        // frame — near the full rate, and since round 3 (five waves per SIMD, four-wide grid nodes, postponed leaves) it is a third faster
        // than the round pipeline on a full 1080p frame.  The round pipeline stands in for trees deeper than the LDS stack columns, then with
        // its exact kernels on (a serial walk of the reference tree, ~1.5 ms, sits on the critical path of every round); chosen explicitly
        // (RTAMD_KERNEL=wavefront) it keeps the padded box test's answer unless RTAMD_ROUNDS_EXACT=1.  RTAMD_AUTO_GROUPS_PER_CU=n: opt
        // into the round pipeline from n sub-tiles per CU on.
        bool use_persistent = use_wavefront && !(ksel && strcmp(ksel, "wavefront") == 0) &&
                              scene->info.bvh_depth <= P8_STACK && scene->light_walk_depth <= P8_STACK && !getenv("RTAMD_WF_LDS_STACK");
        if (use_persistent && !ksel) {
            const uint64_t auto_groups = (uint64_t)(getenv("RTAMD_AUTO_GROUPS_PER_CU") ? atoi(getenv("RTAMD_AUTO_GROUPS_PER_CU")) : 0);
            if (auto_groups && (uint64_t)n_work * (uint64_t)streams >= auto_groups * (uint64_t)scene->n_cus) use_persistent = false;
        }
        const bool hw6_persistent = scene->flavor == RT_INTEGRATOR_HW6 && scene->hw6_pt_stack && !(ksel && strcmp(ksel, "mega") == 0) && !getenv("RTAMD_HW6_SCRATCH_STACK");
        if (streams > 1 && !use_wavefront && !hw6_persistent) return fail(RT_ERR_UNSUPPORTED, "rt_render: sample_streams > 1 needs the persistent / round kernels (RTAMD_KERNEL=mega or a tree beyond their limits is in effect)");
        // throughput mode (include/rtamd.h: sample_streams): K path slots per pixel, `samples` per stream, a partial-sum buffer and a final reduction
        auto setup_streams = [&]() {
            R.streams = streams; R.n_pixslots = n_work * 64u; R.seed_stride = (uint32_t)R.width * (uint32_t)R.height;
            R.total_samples = (uint32_t)R.samples;
            R.sample_seeds = (p->flags & RT_FLAG_SAMPLE_SEEDS) ? 1u : 0u;
            R.rr_depth = (p->flags & RT_FLAG_RUSSIAN_ROULETTE) ? 2 : 0;
            R.samples /= streams;                           // per stream; inv_samples stays 1 / (all samples of the pixel)
            R.sample_stop = R.samples;
            const size_t need = (size_t)streams * R.n_pixslots * 3 * sizeof(float);
            if (scene->partial_bytes < need) {
                if (scene->d_partial) (void)hipFree(scene->d_partial);
                scene->d_partial = nullptr; scene->partial_bytes = 0;
                HIP_CHECK(hipMalloc((void **)&scene->d_partial, need));
                scene->partial_bytes = need;
            }
            R.partial = scene->d_partial;
        };
        if (txt_scene && p->integrator == RT_INTEGRATOR_HW3 && R.ray_depth > RT3_MAX_DEPTH) return fail(RT_ERR_LIMIT, "rt_render: hw3 ray_depth above 8");
        if (txt_scene && scene->txt_has_triangles && p->integrator != RT_INTEGRATOR_HW5) return fail(RT_ERR_INVALID_ARG, "rt_render: a .txt scene with TRIANGLE figures renders with RT_INTEGRATOR_HW5 only");
        if (p->integrator == RT_INTEGRATOR_HW5 && R.ray_depth > RT4_MAX_DEPTH) return fail(RT_ERR_LIMIT, "rt_render: hw5 ray_depth above 8");
        if (p->integrator == RT_INTEGRATOR_HW5 && (scene->info.bvh_depth > RT5_STACK || scene->info.light_bvh_depth > RT5_STACK)) return fail(RT_ERR_LIMIT, "rt_render: hw5 BVH deeper than 64");
        if (p->integrator == RT_INTEGRATOR_HW4 && R.ray_depth > RT4_MAX_DEPTH) return fail(RT_ERR_LIMIT, "rt_render: hw4 ray_depth above 8");
        if (p->integrator == RT_INTEGRATOR_HW4 && scene->viewt.n_light_prims > RT4_MAX_LIGHTS) return fail(RT_ERR_LIMIT, "rt_render: hw4 supports at most 32 emissive box/ellipsoid lights");
        if (p->integrator == RT_INTEGRATOR_HW2 && R.ray_depth > RT2_MAX_DEPTH) return fail(RT_ERR_LIMIT, "rt_render: hw2 ray_depth above 16");
        if (p->integrator == RT_INTEGRATOR_HW1 && R.shard_count > 1) return fail(RT_ERR_UNSUPPORTED, "rt_render: the hw1 caster renders unsharded frames only");
        const bool float_tan = p->integrator == RT_INTEGRATOR_HW1 || p->integrator == RT_INTEGRATOR_HW2;
        const float txt_tan_fov_y = (float_tan ? scene->viewt.tan_fov_x_f : scene->viewt.tan_fov_x) * R.height / R.width; // hw3/src/scene.cpp:101
        if (scene->flavor == RT_INTEGRATOR_HW6 && R.ray_depth > RT6_MAX_DEPTH) return fail(RT_ERR_LIMIT, "rt_render: hw6 ray_depth above 8");
        SceneView V8 = scene->view; // per-render copy: the hw7 replay switches are render parameters, not scene state
        if (hw7) { V8.hw7 = 1; V8.last_level_emission_only = 0; V8.env_image = -1; }
        // Exactness follows the scene, not the pipeline: when the persistent pipeline cannot take the scene (a tree deeper than its
        // stack columns), the round pipeline runs with its exact kernels on.  Only an explicit RTAMD_KERNEL=wavefront (the yardstick
        // of the benchmarks; RTAMD_ROUNDS_EXACT=1 switches the exact kernels on there too) and the megakernel, which has no gate,
        // keep the padded boxes' answer — and say so in rt_stats.reference_exact.
        const bool rounds_chosen = ksel && strcmp(ksel, "wavefront") == 0;
        if (!use_persistent && (!use_wavefront || (rounds_chosen && !getenv("RTAMD_ROUNDS_EXACT")))) V8.exact_boxes = 0;
        if (!V8.exact_boxes) V8.cull_k = 4.8e-7f; // no exact walks to feed: the walkers look behind the best hit by the tie tolerance only
        uint32_t launches = 0;
        bool time_trace = false, use_persistent6 = false;
        HIP_CHECK(hipEventRecord(scene->ev_start, stream));
        if (blocks) {
            if (use_wavefront) {
                if (streams > 1) setup_streams();
                // every traverse launch is bracketed by events when stats are wanted -- up to 64 k rounds (e.g. 10,922 spp at depth 6)
                if (use_persistent) {
                    time_trace = stats != nullptr;
                    launch_persistent(scene, V8, R, n_work * (uint32_t)streams, stream, count, time_trace);
                    launches = scene->pt_launches;
                } else {
                time_trace = stats != nullptr && wavefront_rounds(V8, R) * (size_t)wavefront_pipelines(n_work * (uint32_t)streams) <= 65536;
                launch_wavefront(scene, V8, R, n_work * (uint32_t)streams, stream, count, time_trace);
                launches = (uint32_t)scene->wf_pipes * (1 + 2 * (uint32_t)wavefront_rounds(V8, R));
                }
                if (streams > 1) {
                    hipLaunchKernelGGL(dev::wf_reduce_streams_kernel, dim3((R.n_pixslots + 255u) / 256u), dim3(256), 0, stream, R);
                    HIP_CHECK(hipGetLastError());
                    launches++;
                }
            } else if (p->integrator == RT_INTEGRATOR_HW1) {
                uint32_t npx = (uint32_t)R.width * (uint32_t)R.height;
                hipLaunchKernelGGL(dev::render_hw1_kernel, dim3((npx + 255) / 256), dim3(256), 0, stream, scene->viewt, R.width, R.height, txt_tan_fov_y, d_rgb, d_rgb8);
                HIP_CHECK(hipGetLastError());
                launches = 1;
            } else if (p->integrator == RT_INTEGRATOR_HW5) {
                hipLaunchKernelGGL(dev::render_hw5_kernel, dim3(blocks), dim3(64), 0, stream, scene->view5, R, txt_tan_fov_y, n_work);
                HIP_CHECK(hipGetLastError());
                launches = 1;
            } else if (p->integrator == RT_INTEGRATOR_HW4) {
                hipLaunchKernelGGL(dev::render_hw4_kernel, dim3(blocks), dim3(64), 0, stream, scene->viewt, R, txt_tan_fov_y, n_work);
                HIP_CHECK(hipGetLastError());
                launches = 1;
            } else if (p->integrator == RT_INTEGRATOR_HW2) {
                hipLaunchKernelGGL(dev::render_hw2_kernel, dim3(blocks), dim3(64), 0, stream, scene->viewt, R, txt_tan_fov_y, n_work);
                HIP_CHECK(hipGetLastError());
                launches = 1;
            } else if (txt_scene) {
                hipLaunchKernelGGL(dev::render_hw3_kernel, dim3(blocks), dim3(64), 0, stream, scene->viewt, R, txt_tan_fov_y, n_work);
                HIP_CHECK(hipGetLastError());
                launches = 1;
            } else if (hw6_persistent) {
                // persistent dataflow organisation (device/rt_persistent_hw6.h); RTAMD_KERNEL=mega keeps the per-lane path machine
                use_persistent6 = true;
                time_trace = stats != nullptr;
                if (streams > 1) setup_streams();
                launch_persistent6(scene, scene->view6, R, n_work * (uint32_t)streams, stream, count, time_trace);
                launches = scene->pt_launches;
                if (streams > 1) {
                    hipLaunchKernelGGL(dev::wf_reduce_streams_kernel, dim3((R.n_pixslots + 255u) / 256u), dim3(256), 0, stream, R);
                    HIP_CHECK(hipGetLastError());
                    launches++;
                }
            } else if (scene->flavor == RT_INTEGRATOR_HW6) {
                if (scene->hw6_lds_stack && !getenv("RTAMD_HW6_SCRATCH_STACK")) hipLaunchKernelGGL(dev::render_hw6_kernel<true>, dim3(blocks), dim3(64), 0, stream, scene->view6, R, n_work);
                else hipLaunchKernelGGL(dev::render_hw6_kernel<false>, dim3(blocks), dim3(64), 0, stream, scene->view6, R, n_work);
                HIP_CHECK(hipGetLastError());
                launches = 1;
            } else {
                if (count) hipLaunchKernelGGL(dev::render_hw8_kernel<true>, dim3(blocks), dim3(64), 0, stream, V8, R, n_work);
                else hipLaunchKernelGGL(dev::render_hw8_kernel<false>, dim3(blocks), dim3(64), 0, stream, V8, R, n_work);
                HIP_CHECK(hipGetLastError());
                launches = 1;
            }
        }
        HIP_CHECK(hipEventRecord(scene->ev_stop, stream));
        if (own_rgb) HIP_CHECK(hipMemcpyAsync(out_rgb, d_rgb, elems * sizeof(float), hipMemcpyDeviceToHost, stream));
        if (own_rgb8) HIP_CHECK(hipMemcpyAsync(out_rgb8, d_rgb8, elems, hipMemcpyDeviceToHost, stream));
        unsigned long long h_cnt[64] = {0};
        HIP_CHECK(hipMemcpyAsync(h_cnt, scene->d_counters, 512, hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream)); // render is synchronous on return
        use_persistent = use_persistent && use_wavefront && blocks;
        if (use_persistent6 && getenv("RTAMD_DEBUG_COUNTERS")) {
            fprintf(stderr, "[rtamd] persistent hw6 pipeline: %u launches; re-deal %.2f ms on the host (slowest workgroup / mean under the round-robin deal: %.3f); light sums through the slow role %llu of %llu; exact closest-hit walks %llu of %llu, exact light sums %llu\n",
                    scene->pt_launches, scene->pt_rebalance_ms, scene->pt_imbalance, h_cnt[13], h_cnt[1], h_cnt[12], h_cnt[0], h_cnt[11]);
            if (count) {
                fprintf(stderr, "[rtamd] persistent hw6 kernel, light sums in the slow role by number of hits (0..14, 15+):");
                for (int b = 0; b < 16; b++) fprintf(stderr, " %llu", h_cnt[32 + b]);
                fprintf(stderr, "\n");
                const double tt = (double)(h_cnt[16] + h_cnt[17] + h_cnt[18] + h_cnt[19] + h_cnt[20]);
                fprintf(stderr, "[rtamd] persistent hw6 kernel, wave time by role: closest-hit walks %.1f %%, light walks %.1f %%, shading %.1f %%, slow light sums %.1f %%, idle %.1f %%\n",
                        100 * h_cnt[16] / tt, 100 * h_cnt[17] / tt, 100 * h_cnt[18] / tt, 100 * h_cnt[19] / tt, 100 * h_cnt[20] / tt);
            }
            if (scene->d_pt_debug && scene->pt_blocks <= PT_DEBUG_BLOCKS) {
                std::vector<unsigned long long> dbg((size_t)scene->pt_blocks * 3);
                HIP_CHECK(hipMemcpy(dbg.data(), scene->d_pt_debug, dbg.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
                unsigned long long t0 = ~0ull, tmin = ~0ull, tmax = 0; double tsum = 0;
                for (uint32_t b = 0; b < scene->pt_blocks; b++) if (dbg[3 * b] && dbg[3 * b] < t0) t0 = dbg[3 * b];
                for (uint32_t b = 0; b < scene->pt_blocks; b++) { unsigned long long e = dbg[3 * b + 1] - t0; tmin = e < tmin ? e : tmin; tmax = e > tmax ? e : tmax; tsum += (double)e; }
                fprintf(stderr, "[rtamd] persistent hw6 kernel (last launch): %u workgroups, exit times min / mean / max = %.3f / %.3f / %.3f ms after the first start\n",
                        scene->pt_blocks, tmin * 1e-5, tsum / scene->pt_blocks * 1e-5, tmax * 1e-5);
                if (const char *dump = getenv("RTAMD_DUMP_WG")) { // diagnostic, as for hw8 (tools/tuning/wg_balance.py)
                    if (FILE *f = fopen(dump, "w")) {
                        for (uint32_t b = 0; b < scene->pt_blocks; b++) fprintf(f, "%u %.4f %.4f %llu\n", b, (dbg[3 * b] - t0) * 1e-5, (dbg[3 * b + 1] - t0) * 1e-5, dbg[3 * b + 2]);
                        fclose(f);
                    }
                }
            }
        }
        if (use_persistent6 && h_cnt[29]) return fail(RT_ERR_LIMIT, "rt_render: the persistent hw6 kernel ran into its launch deadline (" + std::to_string(h_cnt[29]) + " waves; RTAMD_PT_TIMEOUT_S raises it); the frame is incomplete");
        if (use_persistent6 && h_cnt[14]) return fail(RT_ERR_HIP, "rt_render: the persistent hw6 kernel lost a path (" + std::to_string(h_cnt[14]) + " waves gave up waiting); the frame is incomplete");
        scene->pipeline = (use_persistent || use_persistent6) ? RT_PIPELINE_PERSISTENT : (use_wavefront && blocks ? RT_PIPELINE_ROUNDS : RT_PIPELINE_SINGLE);
        if (use_persistent) {
            if (h_cnt[29]) return fail(RT_ERR_LIMIT, "rt_render: the persistent kernel ran into its launch deadline (" + std::to_string(h_cnt[29]) + " waves; RTAMD_PT_TIMEOUT_S raises it); the frame is incomplete");
            if (h_cnt[14]) return fail(RT_ERR_HIP, "rt_render: the persistent kernel lost a path (" + std::to_string(h_cnt[14]) + " waves gave up waiting); the frame is incomplete");
            h_cnt[0] -= h_cnt[10] < h_cnt[0] ? h_cnt[10] : h_cnt[0]; // speculative closest-hit queries that the clamp step discarded are not part of the algorithm
            if (getenv("RTAMD_DEBUG_COUNTERS") && scene->d_pt_debug && scene->pt_blocks <= PT_DEBUG_BLOCKS) {
                std::vector<unsigned long long> dbg((size_t)scene->pt_blocks * 3);
                HIP_CHECK(hipMemcpy(dbg.data(), scene->d_pt_debug, dbg.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
                unsigned long long t0 = ~0ull, tmin = ~0ull, tmax = 0; double tsum = 0;
                for (uint32_t b = 0; b < scene->pt_blocks; b++) if (dbg[3 * b] && dbg[3 * b] < t0) t0 = dbg[3 * b];
                for (uint32_t b = 0; b < scene->pt_blocks; b++) { unsigned long long e = dbg[3 * b + 1] - t0; tmin = e < tmin ? e : tmin; tmax = e > tmax ? e : tmax; tsum += (double)e; }
                if (count) {
                    const double tt = (double)(h_cnt[16] + h_cnt[17] + h_cnt[18] + h_cnt[19] + h_cnt[20]);
                    fprintf(stderr, "[rtamd] persistent kernel, wave time by role: closest-hit walks %.1f %%, light walks %.1f %%, shading %.1f %%, exact walks %.2f %%, idle %.1f %%; "
                                    "walker lane utilisation: closest hit %.1f of 64 (%llu wave iterations), light %.1f of 64 (%llu); %llu stints, %llu shade batches of %.1f paths\n",
                            100 * h_cnt[16] / tt, 100 * h_cnt[17] / tt, 100 * h_cnt[18] / tt, 100 * h_cnt[19] / tt, 100 * h_cnt[20] / tt,
                            (double)h_cnt[22] / (double)(h_cnt[21] ? h_cnt[21] : 1), h_cnt[21], (double)h_cnt[24] / (double)(h_cnt[23] ? h_cnt[23] : 1), h_cnt[23],
                            h_cnt[25], h_cnt[26], (double)h_cnt[27] / (double)(h_cnt[26] ? h_cnt[26] : 1));
                    for (int w = 0; w < 2; w++) {
                        const double tw = (double)(h_cnt[48 + 3 * w] + h_cnt[49 + 3 * w] + h_cnt[50 + 3 * w]);
                        fprintf(stderr, "[rtamd]   %s walker's wave time: hand-off and refill %.1f %%, inner nodes %.1f %%, leaves %.1f %%; leaf passes %llu with %.1f of 64 lanes\n",
                                w ? "light" : "closest-hit", 100 * h_cnt[48 + 3 * w] / tw, 100 * h_cnt[49 + 3 * w] / tw, 100 * h_cnt[50 + 3 * w] / tw,
                                h_cnt[54 + 2 * w], (double)h_cnt[55 + 2 * w] / (double)(h_cnt[54 + 2 * w] ? h_cnt[54 + 2 * w] : 1));
                    }
                    fprintf(stderr, "[rtamd]   closest-hit walker's hand-off points: %llu; of their time: publishing finished walks %.1f %%, taking new ones from the bitmap %.1f %%, reading their rays %.1f %% (the rest: the test itself)\n",
                            h_cnt[63], 100.0 * h_cnt[60] / (double)(h_cnt[48] ? h_cnt[48] : 1), 100.0 * h_cnt[61] / (double)(h_cnt[48] ? h_cnt[48] : 1), 100.0 * h_cnt[62] / (double)(h_cnt[48] ? h_cnt[48] : 1));
                    fprintf(stderr, "[rtamd]   light tests %llu (%.2f per light sum), hits %llu (%.2f per light sum); triangle tests of closest-hit walks %llu (%.2f per query)\n",
                            h_cnt[59], (double)h_cnt[59] / (double)(h_cnt[1] ? h_cnt[1] : 1), h_cnt[58], (double)h_cnt[58] / (double)(h_cnt[1] ? h_cnt[1] : 1),
                            h_cnt[3] - h_cnt[59], (double)(h_cnt[3] - h_cnt[59]) / (double)(h_cnt[0] ? h_cnt[0] : 1));
                }
                if (const char *dump = getenv("RTAMD_DUMP_WG")) { // diagnostic: start / exit time (ms after the first start) and paths of every workgroup of the last launch
                    if (FILE *f = fopen(dump, "w")) {
                        for (uint32_t b = 0; b < scene->pt_blocks; b++) fprintf(f, "%u %.4f %.4f %llu\n", b, (dbg[3 * b] - t0) * 1e-5, (dbg[3 * b + 1] - t0) * 1e-5, dbg[3 * b + 2]);
                        fclose(f);
                    }
                }
                fprintf(stderr, "[rtamd] persistent pipeline: %u launches; re-deal of the sub-tiles took %.2f ms on the host (slowest workgroup / mean under the round-robin deal: %.3f)\n",
                        scene->pt_launches, scene->pt_rebalance_ms, scene->pt_imbalance);
                fprintf(stderr, "[rtamd] persistent kernel (last launch): %u workgroups, exit times min / mean / max = %.3f / %.3f / %.3f ms after the first start; exact closest hits %llu, exact light sums %llu of %llu + %llu queries\n",
                        scene->pt_blocks, tmin * 1e-5, tsum / scene->pt_blocks * 1e-5, tmax * 1e-5, h_cnt[12], h_cnt[13], h_cnt[0], h_cnt[1]);
            }
        }
        if (count && use_wavefront && blocks && !use_persistent) { // queries = lengths of the per-round queues
            size_t rounds = wavefront_rounds(V8, R);
            std::vector<uint32_t> ctr(scene->wf_ctr_block * scene->wf_pipes);
            HIP_CHECK(hipMemcpy(ctr.data(), scene->wf.ctr, ctr.size() * 4, hipMemcpyDeviceToHost));
            for (int h = 0; h < scene->wf_pipes; h++)
                for (size_t r = 0; r < rounds; r++) {
                    const uint32_t *c = ctr.data() + (size_t)h * scene->wf_ctr_block + WF_CTR * r;
                    h_cnt[0] += c[0];
                    if (scene->info.n_lights) { h_cnt[1] += c[1]; h_cnt[11] += c[4]; }
                }
            h_cnt[0] -= h_cnt[10]; // speculative closest-hit queries that the clamp step discarded are not part of the algorithm
            h_cnt[13] = h_cnt[11];  // light sums finished by the exact kernel
        }
        if (count && use_wavefront && getenv("RTAMD_DEBUG_COUNTERS")) { // wave iterations a query stays in flight, buckets of 32
            fprintf(stderr, "[rtamd] closest-hit queries by in-flight wave iterations (x32):");
            for (int b = 0; b < 16; b++) fprintf(stderr, " %llu", h_cnt[16 + b]);
            fprintf(stderr, "\n[rtamd] light queries by in-flight wave iterations (x32):");
            for (int b = 0; b < 16; b++) fprintf(stderr, " %llu", h_cnt[32 + b]);
            fprintf(stderr, "\n");
        }
        if (count && getenv("RTAMD_DEBUG_COUNTERS"))
            fprintf(stderr, "[rtamd] light queries finished by the exact kernel: %llu of %llu; trace kernel: wave node-iterations %llu, leaf phases %llu (lanes %llu), refills %llu; lane node visits %llu, tri tests %llu\n",
                    h_cnt[11], h_cnt[1], h_cnt[4], h_cnt[5], h_cnt[6], h_cnt[7], h_cnt[8], h_cnt[9]);
        if (stats) {
            float ms = 0;
            HIP_CHECK(hipEventElapsedTime(&ms, scene->ev_start, scene->ev_stop));
            memset(stats, 0, sizeof *stats);
            stats->kernel_ms = ms;
            stats->total_ms = now_ms() - t0;
            stats->launches = launches;
            stats->pipeline = (uint32_t)scene->pipeline;
            stats->reference_exact = p->integrator == RT_INTEGRATOR_HW5 ? 1u : use_persistent6 ? (scene->view6.exact_boxes ? 1u : 0u)
                                     : ((scene->flavor == RT_INTEGRATOR_HW8 && (use_persistent || (use_wavefront && blocks)) && V8.exact_boxes == 1u) ? 1u : 0u);
            if (use_persistent || use_persistent6) {
                double sum = 0;
                for (uint32_t pp = 0; time_trace && pp < scene->pt_launches; pp++) { float e = 0; HIP_CHECK(hipEventElapsedTime(&e, scene->ev_pool[2 * pp], scene->ev_pool[2 * pp + 1])); sum += e; }
                stats->dominant_kernel_ms = time_trace ? sum : ms; stats->dominant_kernel_launches = scene->pt_launches;
                stats->exact_closest_hits = h_cnt[12]; stats->exact_light_sums = use_persistent6 ? h_cnt[11] : h_cnt[13];
            } else if (use_wavefront && blocks && time_trace) {
                stats->exact_closest_hits = h_cnt[12]; stats->exact_light_sums = h_cnt[13]; // counting renders only
                size_t rounds = wavefront_rounds(V8, R);
                double sum = 0;
                const size_t n_launch = rounds * (size_t)scene->wf_pipes; // with more than one pipeline a launch shares the GPU with the other pipelines' kernels
                for (size_t r = 0; r < n_launch; r++) { float e = 0; HIP_CHECK(hipEventElapsedTime(&e, scene->ev_pool[2 * r], scene->ev_pool[2 * r + 1])); sum += e; }
                stats->dominant_kernel_ms = sum; stats->dominant_kernel_launches = (uint32_t)n_launch;
                if (const char *path = getenv("RTAMD_DUMP_ROUNDS")) { // diagnostic: per launch of the traverse kernel its queue lengths and duration
                    std::vector<uint32_t> ctr(scene->wf_ctr_block * scene->wf_pipes);
                    HIP_CHECK(hipMemcpy(ctr.data(), scene->wf.ctr, ctr.size() * 4, hipMemcpyDeviceToHost));
                    if (FILE *f = fopen(path, "a")) { // appended: one block per render
                        fprintf(f, "round,pipeline,closest_hit_queries,light_queries,traverse_ms\n");
                        for (size_t r = 0; r < rounds; r++)
                            for (int h = 0; h < scene->wf_pipes; h++) {
                                float e = 0; (void)hipEventElapsedTime(&e, scene->ev_pool[2 * (r * scene->wf_pipes + h)], scene->ev_pool[2 * (r * scene->wf_pipes + h) + 1]);
                                const uint32_t *c = ctr.data() + (size_t)h * scene->wf_ctr_block + WF_CTR * r;
                                fprintf(f, "%zu,%d,%u,%u,%.4f\n", r, h, c[0], c[1], e);
                            }
                        fclose(f);
                    }
                }
            } else { stats->dominant_kernel_ms = ms; stats->dominant_kernel_launches = launches; }
            // pixels of this shard that lie inside the image
            uint64_t px = 0;
            for (uint32_t st = 0; st < R.n_shard_tiles; st++) {
                uint32_t gt = R.shard_count > 1 ? (uint32_t)R.shard_index + st * (uint32_t)R.shard_count : st;
                int tx0 = (int)(gt % (uint32_t)R.tiles_x) * R.tile_w, ty0 = (int)(gt / (uint32_t)R.tiles_x) * R.tile_h;
                int w = R.width - tx0 < R.tile_w ? R.width - tx0 : R.tile_w, h = R.height - ty0 < R.tile_h ? R.height - ty0 : R.tile_h;
                px += (uint64_t)w * h;
            }
            stats->samples = px * (uint64_t)R.samples * (uint64_t)streams;
            stats->closest_hit_queries = h_cnt[0]; stats->light_pdf_queries = h_cnt[1];
            stats->node_visits = h_cnt[2]; stats->triangle_tests = h_cnt[3];
        }
        return RT_OK;
    } catch (const HipError &e) {
        return fail(RT_ERR_HIP, e.what());
    } catch (const std::exception &e) { // e.g. std::bad_alloc from the host-side vectors: nothing crosses the C boundary
        return fail(RT_ERR_INVALID_ARG, std::string("rt_render: ") + e.what());
    }
}

int rt_unshard(const rt_render_params *p, const void *shard_buf, size_t elem_size, void *full_image) {
    if (!p || !shard_buf || !full_image || (elem_size != 1 && elem_size != 4)) return fail(RT_ERR_INVALID_ARG, "rt_unshard: bad argument");
    RenderView R{};
    std::string err;
    if (!resolve_tiles(p, R, err)) return fail(RT_ERR_INVALID_ARG, "rt_unshard: " + err);
    const uint8_t *src = (const uint8_t *)shard_buf;
    uint8_t *dst = (uint8_t *)full_image;
    size_t px = 3 * elem_size;
    if (R.shard_count <= 1) { memcpy(dst, src, (size_t)R.width * R.height * px); return RT_OK; }
    for (uint32_t st = 0; st < R.n_shard_tiles; st++) {
        uint32_t gt = (uint32_t)R.shard_index + st * (uint32_t)R.shard_count;
        int tx0 = (int)(gt % (uint32_t)R.tiles_x) * R.tile_w, ty0 = (int)(gt / (uint32_t)R.tiles_x) * R.tile_h;
        int w = R.width - tx0 < R.tile_w ? R.width - tx0 : R.tile_w, h = R.height - ty0 < R.tile_h ? R.height - ty0 : R.tile_h;
        for (int ly = 0; ly < h; ly++)
            memcpy(dst + ((size_t)(ty0 + ly) * R.width + tx0) * px, src + (((size_t)st * R.tile_h + ly) * R.tile_w) * px, (size_t)w * px);
    }
    return RT_OK;
}

// ---- host-side front-end ---------------------------------------------------------------------------
int rt_load_gltf(const char *path, int flavor, rt_host_scene **out) {
    if (!path || !out) return fail(RT_ERR_INVALID_ARG, "rt_load_gltf: null argument");
    *out = nullptr;
    try {
        *out = load_gltf(path, flavor);
        return RT_OK;
    } catch (const std::exception &e) {
        return fail(RT_ERR_PARSE, std::string("rt_load_gltf(") + path + "): " + e.what());
    }
}
int rt_load_txt(const char *path, int flavor, rt_host_scene **out, int32_t *w, int32_t *h, int32_t *samples, int32_t *depth) {
    if (!path || !out) return fail(RT_ERR_INVALID_ARG, "rt_load_txt: null argument");
    *out = nullptr;
    try {
        *out = load_txt(path, flavor, w, h, samples, depth);
        return RT_OK;
    } catch (const std::exception &e) {
        return fail(RT_ERR_PARSE, std::string("rt_load_txt(") + path + "): " + e.what());
    }
}
int rt_host_scene_set_environment(rt_host_scene *hs, const char *image_path) {
    if (!hs || !image_path) return fail(RT_ERR_INVALID_ARG, "rt_host_scene_set_environment: null argument");
    try {
        int w, h;
        load_image_rgb8(image_path, w, h, hs->env_data);
        hs->env = rt_image{w, h, nullptr};
        hs->has_env = true;
        hs->finalize();
        return RT_OK;
    } catch (const std::exception &e) {
        return fail(RT_ERR_IO, e.what());
    }
}
const rt_scene_desc *rt_host_scene_desc(const rt_host_scene *hs) { return hs ? &hs->desc : nullptr; }
void rt_host_scene_free(rt_host_scene *hs) { delete hs; }

int rt_write_ppm(const char *path, int32_t width, int32_t height, const uint8_t *rgb8) { // sceneio.cpp:383-385,397-401
    if (!path || !rgb8 || width <= 0 || height <= 0) return fail(RT_ERR_INVALID_ARG, "rt_write_ppm: bad argument");
    FILE *f = fopen(path, "wb");
    if (!f) return fail(RT_ERR_IO, std::string("rt_write_ppm: cannot open ") + path);
    fprintf(f, "P6\n%d %d\n255\n", width, height);
    size_t n = (size_t)width * height * 3;
    bool ok = fwrite(rgb8, 1, n, f) == n;
    ok = (fclose(f) == 0) && ok;
    return ok ? RT_OK : fail(RT_ERR_IO, std::string("rt_write_ppm: short write to ") + path);
}
int rt_decode_png(const char *path, int32_t *width, int32_t *height, uint8_t **rgb) {
    if (!path || !width || !height || !rgb) return fail(RT_ERR_INVALID_ARG, "rt_decode_png: null argument");
    try {
        std::vector<uint8_t> px;
        int w, h;
        load_image_rgb8(path, w, h, px);
        *rgb = (uint8_t *)malloc(px.size());
        if (!*rgb) return fail(RT_ERR_IO, "rt_decode_png: out of memory");
        memcpy(*rgb, px.data(), px.size());
        *width = w; *height = h;
        return RT_OK;
    } catch (const std::exception &e) {
        return fail(RT_ERR_IO, e.what());
    }
}
void rt_free(void *p) { free(p); }

} // extern "C"
