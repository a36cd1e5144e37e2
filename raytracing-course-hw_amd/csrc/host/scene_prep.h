#pragma once
#include "../../../include/rtamd.h"
#include "../device/rt_types.h"
#include "../device/rt_types_hw6.h"
#include "../device/rt_types_hw5.h"
#include <cstdint>
#include <vector>

namespace rtamd {

struct PreparedScene {
    std::vector<GpuNode> nodes, light_nodes;
    // The reference's own trees with their unpadded boxes, and every figure's own box (BVH order): what the reference-exact
    // walks of the persistent pipeline read (device/rt_persistent.h); box_c2 = 2^-20 * the largest |coordinate| of scene and camera.
    std::vector<GpuRefNode> ref_nodes, ref_light_nodes;
    std::vector<float> tri_box;
    // Per figure the box of its LEAF in the reference's tree (a few leaves hold several triangles): what the GPU builder makes the
    // walkers' tree from, so that whatever the reference can reach through its leaf box is inside a box of ours.  Empty when the
    // reference's tree is not replayed.
    std::vector<float> walk_box;
    // The same for the lights (light order): the box of each light's leaf in the reference's light tree.  The persistent kernel's light
    // walker uses a tree of its own over these (built on the GPU): the reference's light tree also holds the far-apart ceiling quads and
    // the lamp near its root, so its upper boxes span the room and most rays descend into them in vain.
    std::vector<float> light_walk_box;
    // Tripwires (device/rt_exact.h pt_tripwire): the reference's triangle test solves the hit's barycentrics in a fixed projection
    // (primitives.cpp:85-104); for a triangle whose plane (nearly) contains that projection's kernel direction the system is (nearly)
    // singular and the test accepts points anywhere on the plane — a "hit" metres away from the triangle, which the reference finds or not
    // depending on whether its walk enters the triangle's leaf box before a closer hit prunes it.  No culling walk can know that, so a ray
    // that pierces the leaf box of such a triangle anywhere AND passes that triangle's test (the reference would test it if its walk got there) goes
    // to the exact walk at once.  Layout: n_tripwire_groups records of two
    // float4 {lo.xyz, first member | hi.xyz, member count} (indices as raw bits), then the members' boxes {lo.xyz, figure index | hi.xyz, -}; empty when
    // the reference's tree is not replayed.
    std::vector<float> tripwires;
    uint32_t n_tripwire_groups = 0;
    float box_c2 = 0.f;
    float box_pad = 0.f;   // absolute part of the walkers' box padding (scene_prep.cpp pad_box): 2^-18 x the largest |coordinate|
    // Order of the light-pdf additions without walking the reference tree: light_sep[j * n_lights + i] = the shallowest
    // separation depth among the boundaries i .. i + 2^j - 1 (boundary b lies between lights b and b+1 of the reference order;
    // its depth is that of the reference-tree node whose children hold the two lights, or, inside one leaf, a pseudo depth
    // that grows towards the leaf's first light).  A range minimum over it gives the depth at which two hit lights separate.
    std::vector<uint16_t> light_sep;
    uint32_t light_sep_levels = 0;
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
    uint32_t bvh_depth = 0, light_bvh_depth = 0, n_ref_nodes = 0;
};

// Throws std::runtime_error on invalid input.
// tree_on_device (rt_scene_desc.build_flags & RT_BUILD_DEVICE_BVH): the reference's scene-tree build is not replayed; figure order =
// LOAD order, `nodes` stays empty, `isect` / `shade` / `tri_box` stay in LOAD order without leaf marks, for device/rt_bvh_build.h.
void prepare_scene(const rt_scene_desc &desc, PreparedScene &out, bool tree_on_device = false);

// hw6 flavour (flat-shaded triangles, hw6/src/scene.cpp): own scene tree + reference-topology light tree.
struct PreparedScene6 {
    std::vector<GpuNode> nodes, light_nodes, fast_light_nodes;
    std::vector<Tri6> tris, lights, fast_lights;
    uint32_t fast_light_bvh_depth = 0;
    std::vector<uint32_t> light_ref; // 4 words per reference light-tree node: left, right, first, last
    std::vector<uint16_t> light_sep; // as PreparedScene::light_sep, over the reference light tree of hw6
    std::vector<GpuRefNode> ref_nodes, ref_light_nodes; // the reference's own trees with their unpadded boxes, for the exact walks
    std::vector<Tri6> ref_tris;        // figure records in the reference's figure order (ref_index = position)
    std::vector<float> tri_box;        // their own boxes, 8 floats each
    float box_c2 = 0.f;
    std::vector<float> boxes8;       // tree_on_device only: per triangle (LOAD order) the box of its reference leaf
    float box_pad = 0.f;             // as PreparedScene::box_pad
    uint32_t light_sep_levels = 0;
    std::vector<GpuMaterial6> materials;
    std::vector<uint32_t> figure_order, light_order; // reference orders -> LOAD index
    uint32_t bvh_depth = 0, light_bvh_depth = 0, ref_bvh_depth = 0;
};
// tree_on_device: skip the own scene tree; `tris` stay in LOAD order (ref_index set, no leaf marks) and `boxes8` (lo.xyz, -, hi.xyz, -
// per triangle) is filled for device/rt_bvh_build.h.
void prepare_scene_hw6(const rt_scene_desc &desc, PreparedScene6 &out, bool tree_on_device = false);

// hw5 flavour (.txt scene with TRIANGLE figures): the reference's figure order, BVH and light list / light BVH
// (hw5/src/scene.cpp:8-23, hw5/src/include/distributions.h:180-198), both trees in the reference's topology.
struct PreparedScene5 {
    std::vector<GpuNode> nodes, light_nodes;
    std::vector<GpuRefNode> ref_nodes, ref_light_nodes; // the reference's own trees with their unpadded boxes: what the hw5 kernel walks (reference-exact box decisions)
    std::vector<GpuFig5> figs, lights;
    std::vector<uint32_t> figure_order, light_order; // reference orders -> LOAD index
    uint32_t n_nonplanes = 0, bvh_depth = 0, light_bvh_depth = 0;
};
void prepare_scene_hw5(const rt_scene_desc &desc, PreparedScene5 &out);

} // namespace rtamd
