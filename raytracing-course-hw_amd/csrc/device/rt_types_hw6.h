// POD layouts of the hw6 path shared by host preparation and kernels.
#pragma once
#include <stdint.h>
#include "rt_types.h"

namespace rtamd {

// 64-byte figure record for the hw6 triangle test (hw6/src/primitives.cpp:143-164).
struct Tri6 {
    float a[3], b[3], c[3], n[3]; // a = data3, b = data - a, c = data2 - a, n = b.cross(c)
    uint32_t ref_index;           // position in the reference's figure order (tie rule) / light index
    uint32_t last;                // 1 = last record of its leaf
    uint32_t material;
    float point_prob;             // lights: 1 / area (hw6/src/include/distributions.h:121-127)
};
static_assert(sizeof(Tri6) == 64, "Tri6 must be 64 bytes");

struct GpuMaterial6 {
    float color[3]; float ior;
    float emission[3]; int32_t kind; // rt_material_kind
};
static_assert(sizeof(GpuMaterial6) == 32, "GpuMaterial6 must be 32 bytes");

struct SceneView6 {
    const GpuNode *nodes;        // own SAH tree over the figures
    const Tri6 *tris;            // in that tree's leaf order
    const GpuNode *light_nodes;  // reference topology over the light list
    const Tri6 *lights;          // in the reference's light order
    const GpuNode *fast_light_nodes; // own SAH tree over the same lights: finds the hit lights quickly (order-free for <= 2 hits)
    const GpuNode4Q *nodes4, *fast_light_nodes4; // `nodes` / `fast_light_nodes` four wide on the 16-bit grid (rt_types.h): what the persistent pipeline's walkers read
    NodeGrid grid;                   // covers both trees' root boxes and the camera
    const Tri6 *fast_lights;         // the light records in that tree's leaf order (ref_index = position in the reference's light order)
    const uint32_t *light_ref;       // the reference light tree without boxes: 4 words {left, right, first, last} per node (left = 0: leaf)
    const uint16_t *light_sep;       // range-minimum table of the separation depths of neighbouring lights in the reference light tree (scene_prep.h)
    // Reference-exact closest hits (device/rt_exact.h, as for hw8): the reference's own scene tree with its unpadded boxes, the figure
    // records and their boxes (lo.xyz, -, hi.xyz, -) in the reference's figure order
    const GpuRefNode *ref_nodes, *ref_light_nodes;
    const Tri6 *ref_tris;
    const float *tri_box;
    float box_c2, box_c2x, cull_k;            // 2^-20 x the largest |coordinate|; the walkers' relative look-behind
    uint32_t exact_boxes;
    const GpuMaterial6 *materials;
    uint32_t n_tris, n_lights, n_components;
    float n_lights_f, n_components_f; // the same numbers as floats (exact)
    float cam_pos[3], cam_right[3], cam_up[3], cam_fwd[3];
    float bg[3];
    float tan_fov_y;
};

} // namespace rtamd
