// Host side of the on-device scene-tree builder (device/rt_bvh_build.h, rtamd_build.hip).
#pragma once
#include "../device/rt_types.h"
#include <cstdint>
#include <hip/hip_runtime.h>

namespace rtamd {

struct DeviceTree {
    GpuNode *nodes = nullptr;   // hipMalloc'ed, n_nodes two-box nodes, numbered level by level
    uint32_t *order = nullptr;  // hipMalloc'ed, n entries: leaf slot -> primitive (load index)
    uint8_t *last = nullptr;    // hipMalloc'ed, n entries: 1 = last slot of its leaf
    uint32_t n_nodes = 0, depth = 0;
    float build_ms = 0.f;       // GPU time of the build (HIP events)
};

// d_boxes: 8 floats per primitive (lo.xyz, -, hi.xyz, -) on the current device; n >= 1.  Synchronous.  Throws HipError.
// abs_pad: absolute part of the node boxes' padding (scene_prep.cpp pad_box).  max_depth (8..28): no leaf lies deeper, so a traversal
// stack of max_depth entries is enough (the LDS stack columns of the kernel that walks the tree).
DeviceTree build_tree_on_device(const float *d_boxes, uint32_t n, float abs_pad, uint32_t max_depth);
void free_device_tree(DeviceTree &t);

// out[i] = in[order[i]] for records of elem_bytes (a multiple of 16); then the 32-bit word at mark_word_offset of every record becomes
// 1 / 0 (last slot of its leaf or not), or, with or_into_word, keeps its value and gets the mark in bit 0 (mark_word_offset < 0: no marks).
void gather_records(const void *d_in, void *d_out, const DeviceTree &t, uint32_t n, uint32_t elem_bytes, int mark_word_offset, bool or_into_word = false);

// The persistent pipeline's walk nodes on their 16-bit grid (rt_types.h GpuNode4Q, device/rt_node_grid.h):
// The union of the two child boxes of d_nodes[0] (empty children skipped) joins lo / hi.  Synchronous.
void join_root_box(const GpuNode *d_nodes, float lo[3], float hi[3]);

// The tree d_nodes[0, n) with two levels folded into each node (rt_types.h GpuNode4Q), on the grid; hipMalloc'ed.  depth_out: levels of
// the wide tree (at most half the two-box tree's, rounded up).  Synchronous (the fold runs on the host).  Throws when a box does not fit.
GpuNode4Q *widen_nodes(const GpuNode *d_nodes, uint32_t n, const NodeGrid &grid, uint32_t &n_out, uint32_t &depth_out);

} // namespace rtamd
