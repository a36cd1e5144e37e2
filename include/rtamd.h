/*
 * rtamd.h — C-ABI of the MI355X-native path-tracing render path.
 *
 * This is the drop-in boundary for the reference renderer's per-pixel render loop
 * (reference: hw8/src/sceneio.cpp:381-402 `sceneio::renderScene`, whose parallel-for body
 * calls `Scene::getPixel`, hw8/src/scene.cpp:167-177, then the tonemap chain
 * hw8/src/color.cpp:4-31).  The reference has no FFI of its own; the functions below are
 * what a binding for that seam needs (see INTEGRATION.md for the reference-side stub):
 *
 *   rt_scene_create   replaces  Scene::initBVH + Scene::initDistribution  (hw8/src/scene.cpp:65-78)
 *                     plus the upload of the flattened scene to HBM (once).
 *   rt_render         replaces  the `#pragma omp parallel for` of renderScene
 *                     (hw8/src/sceneio.cpp:387-396): per-pixel minstd_rand(y*W+x) replay,
 *                     getPixel, aces_tonemap/gamma_corrected/toExternColorFormat.
 *   rt_scene_destroy  replaces  Scene::~Scene (hw8/src/scene.cpp:56-63).
 *
 * Conventions: plain C, plain pointers and sizes, no exceptions cross the boundary.
 * Every function returns 0 on success and a negative rt_status on failure;
 * rt_last_error() returns a thread-local message for the last failure.  The library owns all
 * device memory; the caller owns every host buffer passed in and may free the rt_scene_desc
 * arrays as soon as rt_scene_create returns.  One rt_scene lives on one GPU (the HIP device
 * current at creation); calls on one rt_scene must be serialised by the caller.
 * There is no CPU fallback: if no HIP device is usable the calls fail with RT_ERR_NO_DEVICE.
 */
#ifndef RTAMD_H
#define RTAMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTAMD_ABI_VERSION 6

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID_ARG = -1,
    RT_ERR_NO_DEVICE = -2,
    RT_ERR_HIP = -3,
    RT_ERR_UNSUPPORTED = -4,
    RT_ERR_IO = -5,
    RT_ERR_PARSE = -6,
    RT_ERR_LIMIT = -7
} rt_status;

/* Which reference integrator the render loop replays. */
typedef enum rt_integrator {
    RT_INTEGRATOR_HW1 = 1, /* ray caster,  hw1/src/scene.cpp:7-30   */
    RT_INTEGRATOR_HW2 = 2, /* Whitted-style: point/directional lights, shadow rays, mirror + Fresnel-weighted refraction, hw2/src/scene.cpp:30-84 */
    RT_INTEGRATOR_HW3 = 3, /* first path tracer over analytic primitives, hw3/src/scene.cpp:31-107 */
    RT_INTEGRATOR_HW4 = 4, /* + cosine / box-light / ellipsoid-light mixture sampling, hw4/src/scene.cpp:10-112, distributions.h */
    RT_INTEGRATOR_HW5 = 5, /* + TRIANGLE primitives, BVH over non-planes, per-pixel engines, hw5/src/scene.cpp:47-112 */
    RT_INTEGRATOR_HW6 = 6, /* triangles, DIFFUSE/METALLIC/DIELECTRIC, hw6/src/scene.cpp:47-105    */
    RT_INTEGRATOR_HW7 = 7, /* hw8's integrator without textures: per-material BRDF without the v.n / l.n gates, alpha = roughness^2,
                              geometric normal in the light pdf (hw7/src/scene.cpp:29-61); renders scenes created for HW8 */
    RT_INTEGRATOR_HW8 = 8  /* glTF PBR + textures + mixture sampling, hw8/src/scene.cpp:84-165     */
} rt_integrator;

/* hw3/hw6 material classes (hw3/src/include/primitives.h, hw6/src/include/primitives.h). */
typedef enum rt_material_kind { RT_MAT_DIFFUSE = 0, RT_MAT_METALLIC = 1, RT_MAT_DIELECTRIC = 2 } rt_material_kind;

/* glTF material as the reference keeps it (hw8/src/include/gltf_structs.h:38-47).
 * Texture fields are glTF *texture* indices (into rt_scene_desc.texture_source) or -1. */
typedef struct rt_material {
    float base_color[3];   /* baseColorFactor rgb, default 1,1,1 */
    float emission[3];     /* emissiveFactor * KHR_materials_emissive_strength, default 0 */
    float metallic_factor; /* default 1 */
    float roughness_factor;/* default 1 */
    int32_t base_color_texture;
    int32_t emissive_texture;
    int32_t metallic_roughness_texture;
    int32_t normal_texture;
    int32_t kind;          /* rt_material_kind; used by the HW3/HW6 integrators only */
    float ior;             /* HW3/HW6 dielectric index of refraction */
} rt_material;

/* 8-bit RGB image, row 0 first, tightly packed (what stbi_load(...,3) hands the reference,
 * hw8/src/sceneio.cpp:374-379). */
typedef struct rt_image {
    int32_t width, height;
    const uint8_t *rgb;
} rt_image;

/* Analytic primitive of the .txt scenes (hw1/hw3: hw3/src/include/primitives.h:17-44). */
typedef enum rt_primitive_type { RT_PRIM_ELLIPSOID = 0, RT_PRIM_PLANE = 1, RT_PRIM_BOX = 2, RT_PRIM_TRIANGLE = 3 } rt_primitive_type;
typedef struct rt_primitive {
    int32_t type;        /* rt_primitive_type */
    float data[3];       /* radii | plane normal | box half-sizes | triangle: Figure::data (third vertex of the TRIANGLE line) */
    float position[3];
    float rotation[4];   /* quaternion x,y,z,w exactly as parsed (may be non-unit) */
    float color[3];
    float emission[3];
    int32_t kind;        /* rt_material_kind */
    float ior;
    float data2[3];      /* TRIANGLE (hw5/src/sceneio.cpp:27-29): second vertex of the line */
    float data3[3];      /* TRIANGLE: first vertex of the line (the BVH / light code's "a") */
} rt_primitive;

/* hw2 light source (hw2/src/include/light_source.h:15-30). */
typedef enum rt_light_type { RT_LIGHT_POINT = 0, RT_LIGHT_DIRECTIONAL = 1 } rt_light_type;
typedef struct rt_light {
    int32_t type;         /* rt_light_type */
    float intensity[3];
    float position[3];    /* point light */
    float attenuation[3]; /* point light: c0 + c1*r + c2*r^2 */
    float direction[3];   /* directional light, as parsed (normalised per query like the reference) */
} rt_light;

typedef struct rt_camera {
    float position[3], right[3], up[3], forward[3];
    float fov_y;         /* glTF scenes (hw6-hw8): yfov in radians */
    float fov_x;         /* .txt scenes (hw1-hw5): CAMERA_FOV_X in radians */
} rt_camera;

/* Scene description: host arrays in LOAD order (the order the reference's loader appends
 * Figures, hw8/src/sceneio.cpp:247-310).  Per triangle the three vertices are stored in the
 * order of the reference's Figure members (data, data2, data3) — i.e. glTF corners (1,3,2),
 * hw8/src/sceneio.cpp:289. */
typedef struct rt_scene_desc {
    uint32_t struct_size;            /* = sizeof(rt_scene_desc) */
    uint32_t n_triangles;            /* hw7 / hw8 scenes: at most 2^24 - 2 (24 bits of figure index in a hit record) */
    const float *positions;          /* n_triangles * 9  */
    const float *texcoords;          /* n_triangles * 6  (may be NULL for HW6) */
    const float *normals;            /* n_triangles * 9  (may be NULL for HW6) */
    const float *tangents;           /* n_triangles * 12 (xyz,w; may be NULL for HW6) */
    const uint32_t *material_index;  /* n_triangles */
    uint32_t n_materials;
    const rt_material *materials;
    uint32_t n_textures;
    const uint32_t *texture_source;  /* glTF textures[i].source */
    uint32_t n_images;
    const rt_image *images;
    const rt_image *environment_map; /* NULL = none (hw8/src/scene.cpp:90-97) */
    uint32_t n_primitives;           /* analytic primitives (HW1-HW5); a scene with neither triangles nor primitives is a .txt
                                        scene when camera.fov_y == 0 && camera.fov_x != 0, else a glTF scene */
    const rt_primitive *primitives;
    rt_camera camera;
    float bg_color[3];
    uint32_t n_lights;               /* HW2 only */
    const rt_light *lights;
    float ambient_light[3];          /* HW2 only (AMBIENT_LIGHT) */
    uint32_t build_flags;            /* RT_BUILD_* */
    uint32_t reserved;
} rt_scene_desc;

/* rt_scene_desc.build_flags (glTF scenes with per-vertex normals, i.e. the hw7 / hw8 integrators):
 * RT_BUILD_DEVICE_BVH  build the scene tree on the GPU (binned SAH, device/rt_bvh_build.h: milliseconds instead of the host's replay of
 *                      the reference's builder, hw8/src/include/bvh.h:34-109) and take the LOAD order as the figure order.  Closest hits
 *                      are the same triangles; what changes is everything the reference's figure order decides: which of two hits at
 *                      exactly equal distance wins, the numbering of the emissive triangles (int(u * N) picks a different light for the
 *                      same random number) and the order of the light-pdf additions.  Frames of such a scene follow the reference's
 *                      estimator, not its pixels -- the contract of throughput mode (rt_render_params.sample_streams).
 *                      hw6 scenes (no normals) always get their tree this way: hw6's own tree never was the reference's, the figure
 *                      order is still replayed on the host and the pixels are the reference's (RTAMD_HOST_BVH=1: host-built tree). */
#define RT_BUILD_DEVICE_BVH 1u

#define RT_FLAG_OUT_DEVICE 1u /* out_rgb_linear / out_rgb8 are device pointers on the scene's GPU */
#define RT_FLAG_COUNTERS   2u /* also fill the work counters of rt_stats (slower kernel variant) */
/* Throughput mode only (sample_streams > 1; not the reference's estimator any more, statistical parity only):
 * RT_FLAG_SAMPLE_SEEDS     every camera sample draws from its own engine, seeded with a hash of (pixel, sample index), instead of
 *                          continuing its stream's engine: samples of a pixel are then independent of how they are dealt to streams;
 * RT_FLAG_RUSSIAN_ROULETTE from the third bounce on a path survives a bounce with probability q = clamp(max component of the
 *                          path's accumulated throughput up to and including this bounce, 0.25, 1) and the bounce's factor is divided
 *                          by q (the reference ends paths by depth and its clamp hack only, hw8/src/scene.cpp:85-87,161-163; both
 *                          stay in force).
 * Both are RT_INTEGRATOR_HW8 / HW7 only (hw6 has the streams, not these options). */
#define RT_FLAG_SAMPLE_SEEDS     4u
#define RT_FLAG_RUSSIAN_ROULETTE 8u

typedef struct rt_render_params {
    uint32_t struct_size; /* = sizeof(rt_render_params) */
    int32_t width, height, samples;
    int32_t ray_depth;    /* 0 = reference default (6, hw8/src/include/scene.h:32) */
    int32_t integrator;   /* rt_integrator */
    /* Sharding for multi-GPU: the image is cut into tile_w x tile_h tiles numbered row-major;
     * tile t belongs to shard (t % shard_count).  shard_count <= 1 renders the whole image
     * into a plain W*H*3 row-major buffer; otherwise the output is the compact sequence of
     * this shard's tiles (each tile_h*tile_w*3, border tiles zero-padded). */
    int32_t tile_w, tile_h;
    int32_t shard_index, shard_count;
    uint32_t flags;
    void *stream;         /* hipStream_t to launch on, NULL = default stream */
    /* 0 or 1 = replay mode: one std::minstd_rand per pixel seeded y*W+x draws all of the pixel's samples, as the reference does
     * (hw8/src/sceneio.cpp:389-391) -- the mode every parity claim is made in.
     * K > 1 = throughput mode (RT_INTEGRATOR_HW8 / HW7 / HW6): K independent streams per pixel, stream k seeded
     * y*W+x + k*W*H and drawing samples/K samples (samples must be a multiple of K, W*H*K < 2^31-1); the same estimator
     * with decorrelated sub-streams, so a small frame or shard fills the GPU.  Deterministic, but NOT the reference's pixels:
     * it agrees with replay mode statistically only. */
    int32_t sample_streams;
    int32_t reserved;     /* must be 0 */
} rt_render_params;

typedef struct rt_stats {
    double kernel_ms;       /* HIP-event time of the render kernel(s) on the launch stream */
    double total_ms;        /* host wall time of rt_render */
    uint64_t samples;       /* camera samples rendered by this call */
    uint64_t closest_hit_queries, light_pdf_queries; /* RT_FLAG_COUNTERS */
    uint64_t node_visits, triangle_tests;            /* RT_FLAG_COUNTERS */
    uint32_t launches;
    uint32_t dominant_kernel_launches; /* launches of the dominant kernel (rt_stats.pipeline says which) */
    double dominant_kernel_ms;         /* sum of their HIP-event durations on the launch stream */
    uint32_t pipeline;                 /* rt_pipeline: how the kernels of this render were organised */
    uint32_t reference_exact;          /* hw6 / hw7 / hw8 renders: 1 = every box decision of this render was the reference's own (the walkers'
                                          hits went through the exactness gate, the doubtful ones through the reference-exact walk);
                                          0 = the answer of the walkers' padded boxes stood (RTAMD_NO_EXACT_BOXES, RT_BUILD_DEVICE_BVH,
                                          RTAMD_KERNEL=wavefront|mega, or a tree beyond the gated pipelines' limits): a pixel in ~1e5 may
                                          then differ from the reference's.  hw5: always 1 (its kernel walks the reference's trees with the reference's own box
                                          test).  0 for hw1..hw4 (no boxes: nothing to decide). */
    uint64_t exact_closest_hits, exact_light_sums; /* RT_PIPELINE_PERSISTENT: queries re-walked with the reference's own box
                                                      arithmetic (hw8/src/primitives.cpp:29-53,163-165) because the fast walk's
                                                      answer was not robust against it */
} rt_stats;

/* Kernel organisations of the hw8 / hw7 integrators (the others are always RT_PIPELINE_SINGLE).  Environment RTAMD_KERNEL =
 * persistent (default) | wavefront | mega selects; all three replay the same arithmetic. */
typedef enum rt_pipeline {
    RT_PIPELINE_SINGLE = 0,     /* one launch of a whole-path kernel: render_hw8_kernel ("mega") and render_hw1..hw6_kernel */
    RT_PIPELINE_ROUNDS = 1,     /* per bounce round: wf_traverse_kernel (dominant) + wf_shade_kernel */
    RT_PIPELINE_PERSISTENT = 2  /* pt_persistent_kernel: one launch per frame, per-path dataflow inside each workgroup */
} rt_pipeline;

typedef struct rt_scene rt_scene;

int rt_abi_version(void);
const char *rt_last_error(void);

int rt_scene_create(const rt_scene_desc *desc, rt_scene **out);
void rt_scene_destroy(rt_scene *scene);

/* Number of floats (or bytes for rgb8) rt_render writes for these params. */
size_t rt_output_elems(const rt_render_params *params);

int rt_render(rt_scene *scene, const rt_render_params *params,
              float *out_rgb_linear /* nullable */, uint8_t *out_rgb8 /* nullable */,
              rt_stats *stats /* nullable */);

/* Scatter a compact shard buffer (layout above) into a full W*H*3 host image. */
int rt_unshard(const rt_render_params *params, const void *shard_buf, size_t elem_size,
               void *full_image);

/* ---- several GPUs in one process ---------------------------------------------------------------------------------------
 * The reference shards nothing (one OpenMP loop over the pixels, hw8/src/sceneio.cpp:387-396, driven from main,
 * hw8/src/main.cpp:7-18); pixels are independent and seeded by their global index, so the frame's 32x32 tiles are dealt
 * round-robin to the devices, every device renders its shard from its own host thread, and ONE exchange step brings the shards
 * to the first device (hipMemcpyPeerAsync over xGMI) where the tiles are scattered into the frame.  The result is bit-identical
 * to the one-device render.  `devices` = HIP device indices (NULL = 0 .. n_devices-1; an index may repeat, which renders two
 * shards on one GPU — used by the tests on a one-GPU machine).  rt_multi_render takes the params of an UNSHARDED frame
 * (shard_count 0 or 1); out pointers are host memory, or memory on the first device with RT_FLAG_OUT_DEVICE; stats are summed
 * over the devices except kernel_ms / dominant_kernel_ms (the slowest device: they render side by side) and total_ms (wall). */
typedef struct rt_multi rt_multi;
int rt_device_count(void);
int rt_multi_create(const rt_scene_desc *desc, const int *devices, int n_devices, rt_multi **out);
int rt_multi_render(rt_multi *multi, const rt_render_params *params, float *out_rgb_linear /* nullable */, uint8_t *out_rgb8 /* nullable */,
                    rt_stats *stats /* nullable */);
void rt_multi_destroy(rt_multi *multi);

/* Scene info the host side needs after preparation. */
typedef struct rt_scene_info {
    uint32_t n_triangles, n_lights, n_bvh_nodes, n_light_bvh_nodes;
    uint32_t bvh_depth, light_bvh_depth;
    uint64_t device_bytes;
    double prep_ms, upload_ms;   /* host preparation; upload (+ the tree build on the GPU when bvh_on_device) */
    double bvh_build_ms;         /* GPU time of the on-device scene-tree build (device/rt_bvh_build.h), 0 when built on the host */
    uint32_t bvh_on_device;      /* 1 = the traversal tree of the scene was built on the GPU */
    uint32_t reserved;
} rt_scene_info;
int rt_scene_get_info(const rt_scene *scene, rt_scene_info *info);
/* Light order chosen by preparation (indices into the LOAD-order triangle arrays);
 * parity-critical: reference hw8/src/include/distributions.h:103-115. */
int rt_scene_get_light_order(const rt_scene *scene, uint32_t *out, uint32_t capacity);

/* Host-only part of rt_scene_create (no GPU needed): the figure order after the reference's BVH build and the light order,
 * as indices into the LOAD-order arrays, for the integrator family the desc belongs to (`integrator` = HW8 / HW6 for glTF
 * scenes, HW5 for .txt scenes).  Either output may be NULL; capacities in elements.  Returns the number of lights. */
int rt_host_prepare_orders(const rt_scene_desc *desc, int integrator, uint32_t *figure_order, uint32_t figure_capacity,
                           uint32_t *light_order, uint32_t light_capacity);

/* ---- host-side front-end (replaces sceneio::loadScene / loadTexture) -------------------- */
typedef struct rt_host_scene rt_host_scene; /* owns the arrays a desc points into */

/* glTF 2.0 subset loader, float-for-float the reference's (hw8/src/sceneio.cpp:348-372).
 * flavor: RT_INTEGRATOR_HW6 or RT_INTEGRATOR_HW8 (material interpretation differs). */
int rt_load_gltf(const char *path, int flavor, rt_host_scene **out);
/* .txt scene loader (flavor RT_INTEGRATOR_HW1..HW5 picks that snapshot's grammar: hw1/src/sceneio.cpp:8-97,
 * hw2/src/sceneio.cpp:8-147, hw3/src/sceneio.cpp:8-107, hw5/src/sceneio.cpp:8-101). Fills width/height/samples/ray_depth. */
int rt_load_txt(const char *path, int flavor, rt_host_scene **out,
                int32_t *width, int32_t *height, int32_t *samples, int32_t *ray_depth);
int rt_host_scene_set_environment(rt_host_scene *hs, const char *image_path);
const rt_scene_desc *rt_host_scene_desc(const rt_host_scene *hs);
void rt_host_scene_free(rt_host_scene *hs);
/* Binary PPM writer (hw8/src/sceneio.cpp:383-385,397-401). */
int rt_write_ppm(const char *path, int32_t width, int32_t height, const uint8_t *rgb8);
/* Image file -> 3-channel RGB8 like stbi_load(path, ..., 3): PNG (8-bit gray/RGB/RGBA/palette, non-interlaced) or
 * baseline JPEG (grayscale / YCbCr, 1x or 2x chroma subsampling), chosen by magic bytes; caller frees with rt_free. */
int rt_decode_png(const char *path, int32_t *width, int32_t *height, uint8_t **rgb);
void rt_free(void *p);

#ifdef __cplusplus
}
#endif
#endif /* RTAMD_H */
