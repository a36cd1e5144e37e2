// POD layouts of the hw5 path (.txt scenes with TRIANGLE figures, a BVH over the non-plane figures and box / ellipsoid /
// triangle lights) shared by host preparation and kernels.
#pragma once
#include <stdint.h>
#include "rt_types.h"

namespace rtamd {

// One figure of hw5/src/include/primitives.h:37-61, 112 bytes = seven float4 loads.
struct GpuFig5 {
    float data[3]; int32_t type;      // rt_primitive_type; data = radii | plane normal | box half-sizes | triangle Figure::data
    float position[3]; int32_t kind;  // rt_material_kind
    float rotation[4];
    float color[3]; float ior;
    float emission[3]; uint32_t last; // 1 = last record of its BVH leaf
    float data2[3]; float pad0;       // triangle: Figure::data2
    float data3[3]; float pad1;       // triangle: Figure::data3 (vertex "a")
};
static_assert(sizeof(GpuFig5) == 112, "GpuFig5 must be 112 bytes");

struct SceneView5 {
    const GpuNode *nodes;        // reference topology over figs[0, n_nonplanes)
    const GpuFig5 *figs;         // the reference's figure order after Scene::initBVH: BVH figures, then the planes
    const GpuNode *light_nodes;  // reference topology over the light list (fixes the order of the pdf additions)
    const GpuRefNode *ref_nodes, *ref_light_nodes; // the same two trees with the reference's UNPADDED boxes: what the kernel walks, with the
                                 // reference's own box test and pruning rule (hw5/src/include/bvh.h:111-141, primitives.cpp:92-116,221-223)
    const GpuFig5 *lights;       // FiguresMix::figures_ order
    uint32_t n_figs, n_nonplanes, n_lights;
    float cam_pos[3], cam_right[3], cam_up[3], cam_fwd[3];
    float bg[3];
    float tan_fov_x;             // (float)tan((double)(fovX / 2)), hw5/src/scene.cpp:118
};

} // namespace rtamd
