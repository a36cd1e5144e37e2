// Flat HBM layouts shared by the host preparation (scene_prep.cpp) and the HIP kernels.
// Everything is plain old data, 16-byte aligned records so one lane fetches a record with
// global_load_dwordx4 instructions.
#pragma once
#include <stdint.h>

namespace rtamd {

// Binary BVH node with BOTH child boxes inline (one 64-byte fetch decides both children).
// child = index of an inner node, or 0x80000000|first for a leaf whose primitives run from `first`
// up to the record whose `pad` word is 1 (cnt repeats the count for diagnostics), or 0xFFFFFFFF (empty).
// Boxes are padded conservatively on the host (see scene_prep.cpp pad_box).
struct GpuNode {
    float lo0[3]; int32_t child0;
    float hi0[3]; int32_t cnt0;
    float lo1[3]; int32_t child1;
    float hi1[3]; int32_t cnt1;
};
static_assert(sizeof(GpuNode) == 64, "GpuNode must be 64 bytes");

// Walk node of the persistent pipeline: the children of a GpuNode's two children (a child that is a leaf stays as it is), each as one
// 16-byte record {x, y, z, child}: two levels of the tree per fetch of one 64-byte line — half the dependent node fetches of a walk.
// The boxes lie on a 16-bit grid over the scene (NodeGrid), rounded outwards and one more cell: x = lo | hi << 16 along x, and so on.
// The walkers move the ray into grid coordinates once per walk; a box of the grid contains the float box whatever that arithmetic
// rounds (rt_device.h slab_test_q).  Unused records: child = 0xFFFFFFFF, a point in the grid's border.
struct GpuNode4Q { uint32_t rec[4][4]; };
static_assert(sizeof(GpuNode4Q) == 64, "GpuNode4Q must be 64 bytes");
#define RT_GRID_CELLS 65000.0   // cells across the larger of: the axis' own extent, 1/64 of the largest extent
#define RT_GRID_BORDER 4        // cells in front of the lowest box
struct NodeGrid { float lo[3], step[3], istep[3]; };

// Per-triangle intersection record, in the reference's BVH (figure) order.  All derived values are
// computed on the host with the reference's float expressions (hw8/src/primitives.cpp:85-104), so
// the kernel's test is bit-identical to Figure::intersectAsTriangle while fetching only 48 bytes.
struct TriIsect {
    float ax, ay, az;   // a = data3.coords
    float nx, ny, nz;   // n = b.cross(c) (negated cross convention), b = data - a, c = data2 - a
    float a1, b1;       // magic1 . b, magic1 . c
    float a2, b2;       // magic2 . b, magic2 . c
    float den;          // b1*a2 - a1*b2
    uint32_t pad;       // triangle arrays: bit 0 = last record of its leaf, bits 1.. = the triangle's index in the figure order (what a hit
                        // reports and the tie rule compares); light arrays: 1 = last record of its leaf
};
static_assert(sizeof(TriIsect) == 48, "TriIsect must be 48 bytes");

// Per-triangle shading record (fetched once per hit): interpolation bases and deltas
// (hw8/src/primitives.cpp:110-117) plus material.
struct TriShade {
    float n3[3], dn1[3], dn2[3];   // data3.normals, data.normals-data3.normals, data2.normals-data3.normals
    float t3[3], dt1[3], dt2[3];   // same for tangents.v
    float uv3[2], duv1[2], duv2[2];
    float tanw;                    // data.tangents.w
    uint32_t material;
    uint32_t orig;                 // LOAD-order index (diagnostics)
    uint32_t pad;
};
static_assert(sizeof(TriShade) == 112, "TriShade must be 112 bytes");

// Emissive triangle in the reference's light order (hw8/src/include/distributions.h:60-115).
struct LightRec {
    TriIsect isect;                // for pdfOneFigureLight's intersection test
    float b[3], c[3];              // for TriangleLight::sample: point = a + u*b + v*c
    float point_prob;              // 1 / area
    float n3[3], dn1[3], dn2[3];   // shading-normal interpolation (pdfOne uses the shading normal in hw8)
    float box_lo[3], pad0;         // the light's own box: min / max over a, a + b, a + c in float (what the robustness test of rt_exact.h compares a hit
    float box_hi[3], pad1;         // point with; formed on the host instead of 24 instructions per light hit)
};
static_assert(sizeof(LightRec) == 48 + 64 + 32, "LightRec must be 144 bytes");

// Node of the reference's own tree with its UNPADDED box (hw8/src/include/bvh.h:9-16), for the reference-exact walks of
// rt_persistent.h: left == 0 marks a leaf holding the figures (lights) [first, last) of the reference order.
struct GpuRefNode {
    float mn[3]; uint32_t left;
    float mx[3]; uint32_t right;
    uint32_t first, last, pad0, pad1;
};
static_assert(sizeof(GpuRefNode) == 48, "GpuRefNode must be 48 bytes");

struct GpuMaterial {
    float base_color[3]; float metallic_factor;
    float emission[3];   float roughness_factor;
    int32_t base_color_tex, emissive_tex, metallic_roughness_tex, normal_tex; // image slot or -1
};
static_assert(sizeof(GpuMaterial) == 48, "GpuMaterial must be 48 bytes");

struct GpuImage {
    uint64_t offset; // byte offset of the RGB8 data in the texel buffer
    int32_t width, height;
};

// Device-side view of a prepared scene (pointers into HBM).
struct SceneView {
    const GpuNode *nodes;          // scene BVH, root = 0
    const TriIsect *tri_isect;     // figure order (the reference's BVH order; LOAD order under RT_BUILD_DEVICE_BVH): shading, exact walks
    float cull_k;                  // walkers prune boxes against best_t (1 + cull_k): the tie tolerance of the leaf test plus the slab test's rounding
    const TriIsect *tri_walk;      // the same records in the leaf order of `nodes` (== tri_isect when `nodes` is the reference topology)
    const TriShade *tri_shade;
    const GpuNode *light_nodes;    // light BVH in the reference's topology, root = 0
    const LightRec *lights;
    // The persistent kernel's light walker: a tree of the library's own over the lights (GPU-built over their reference leaf boxes) and the light
    // records in ITS leaf order, pad = light index (position in `lights`) << 1 | last-of-leaf.  == light_nodes / lights with pads rewritten when the
    // scene has too few lights to bother (then the index is the position).
    const GpuNode *light_walk_nodes;
    const float *tripwires;        // host/scene_prep.h: boxes a ray must not pierce unnoticed (8 floats per record: groups, then members); rt_exact.h pt_tripwire
    uint32_t n_tripwire_groups;
    const GpuNode4Q *nodes4, *light_walk_nodes4;  // `nodes` / `light_walk_nodes` four wide on the 16-bit grid: what the persistent pipeline's walkers read
    NodeGrid grid;                 // covers both trees' root boxes and the camera (every ray origin lies inside it)
    const LightRec *lights_walk;
    const uint16_t *light_sep;     // range-minimum table of the separation depths of neighbouring lights (scene_prep.h)
    const GpuMaterial *materials;
    const GpuImage *images;
    const uint8_t *texels;
    const float *srgb_lut;         // 256 entries: powf(float(1/255.)*b, 2.2f) evaluated by the host libm
    // reference-exact box decisions (rt_persistent.h): the reference's trees with unpadded boxes, per figure its own box
    // (min, max — primitives.cpp:130-141), and the robustness margin c2 = 2^-20 * max |coordinate| of scene and camera
    const GpuRefNode *ref_nodes, *ref_light_nodes;
    const float *tri_box;          // 8 floats per figure: min.xyz, 0, max.xyz, 0
    float box_c2;
    float box_c2x;                 // 1.25f * box_c2 (the walkers' absolute look-behind, rt_exact.h)
    uint32_t exact_boxes;          // 0 = accept every hit of the conservative walk (the round pipeline's behaviour)
    uint32_t n_tris, n_lights, n_components;
    float n_lights_f, n_components_f; // the same numbers as floats (exact): a conversion in the kernel would sit in a VGPR for the whole launch
    uint32_t n_nodes;              // inner nodes of the scene BVH (`nodes`)
    // 1 when every material has 0 <= metallicFactor <= 1 and a non-negative base colour: then the BRDF is >= 0,
    // the throughput of the deepest level is in [0, inf] or NaN, and that level returns exactly its emission
    // (emission + mult*0, or emission via the clamp) — the wavefront path then skips its BRDF/pdf work.
    uint32_t last_level_emission_only;
    // 1 = replay the hw7 snapshot on this scene (set per render): no textures / normal map / environment, alpha =
    // roughness^2 without the 0.08 floor, the ungated BRDF of hw7/src/include/material.h:44-61, and the GEOMETRIC normal in
    // the light pdf (hw7/src/include/distributions.h:140-145).
    uint32_t hw7;
    int32_t env_image;             // image slot of the environment map or -1
    float cam_pos[3], cam_right[3], cam_up[3], cam_fwd[3];
    float bg[3];
    float tan_fov_y;               // (float)tan((double)(fovY / 2)), hw8/src/scene.cpp:180
};

struct RenderView {
    int32_t width, height, samples, ray_depth;
    int32_t sample_stop;           // a path that reaches this sample index parks (its camera ray is in its record): the persistent
                                   // pipeline renders a frame in phases and re-deals the pixels between them; = samples otherwise
    int32_t tile_w, tile_h, tiles_x, tiles_y;
    int32_t shard_index, shard_count;
    uint32_t n_shard_tiles;
    float tan_fov_x;               // tanFovY * width / height, hw8/src/scene.cpp:181
    float inv_samples;             // (float)(1.0 / samples), hw8/src/scene.cpp:176
    float *out_rgb;                // nullable
    uint8_t *out_rgb8;             // nullable
    uint32_t *work_counter;        // dynamic tile queue head
    unsigned long long *counters;  // nullable: [closest, lightq, node_visits, tri_tests]
    // Throughput mode (rt_render_params.sample_streams = K > 1, wavefront path only): K independent random streams per pixel,
    // `samples` is then the count PER STREAM, path slot = stream * n_pixslots + pixel slot, every slot leaves its unnormalised
    // sum in `partial` and wf_reduce_streams_kernel adds the K sums of a pixel in stream order.
    int32_t streams;               // 0 or 1 = replay mode (the reference's one stream per pixel)
    uint32_t n_pixslots;           // pixel slots of this shard (64 per 8x8 sub-tile)
    uint32_t seed_stride;          // stream k of pixel i is seeded with i + k * seed_stride (= width * height)
    float *partial;                // [streams][n_pixslots][3]
    uint32_t sample_seeds;         // throughput mode, RT_FLAG_SAMPLE_SEEDS: every sample seeds its own engine (hash of pixel and sample)
    uint32_t total_samples;        // samples per pixel over all its streams
    int32_t rr_depth;              // throughput mode, RT_FLAG_RUSSIAN_ROULETTE: first bounce index that plays roulette (0 = off)
};

} // namespace rtamd
