#pragma once
#include "../../../include/rtamd.h"
#include "../device/rt_types.h"
#include <cstdint>
#include <vector>

namespace rtamd {

struct PreparedScene {
    std::vector<GpuNode> nodes, light_nodes;
    std::vector<TriIsect> isect;
    std::vector<TriShade> shade;
    std::vector<LightRec> lights;
    std::vector<GpuMaterial> materials;
    std::vector<GpuImage> images;
    std::vector<uint8_t> texels;
    float srgb_lut[256];
    int32_t env_image = -1;
    std::vector<uint32_t> figure_order; // BVH order -> LOAD index
    std::vector<uint32_t> light_order;  // light order -> LOAD index
    uint32_t bvh_depth = 0, light_bvh_depth = 0, ref_nodes = 0;
};

// Throws std::runtime_error on invalid input.
void prepare_scene(const rt_scene_desc &desc, PreparedScene &out);

} // namespace rtamd
