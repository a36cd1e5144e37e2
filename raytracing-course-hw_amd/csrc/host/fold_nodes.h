// The fold of a two-box tree into the persistent pipeline's four-wide grid nodes (rt_types.h GpuNode4Q), on host vectors: shared by
// rtamd_build.hip (widen_nodes: download, fold, upload) and the test hooks (tests/test_fold_nodes.py runs it without a GPU).
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>
#include "../device/rt_node_grid.h"

namespace rtamd {

// Every wide node takes the children of a two-box node's two children (a child that is a leaf stays as it is; empty children vanish), so a
// wide tree has at most half the levels (rounded up).  Wide nodes are numbered in breadth-first order: the four children of a node's
// record k that are inner nodes follow each other.  Throws when a box does not fit the grid or a child index is out of range.
inline void fold_nodes(const std::vector<GpuNode> &nodes, const NodeGrid &grid, std::vector<GpuNode4Q> &wide, uint32_t &depth_out) {
    struct Entry { const float *lo, *hi; uint32_t child; };
    std::vector<uint32_t> source{0u}, level{1u}; // wide node -> the two-box node it folds, its level
    wide.clear();
    wide.reserve(nodes.size() / 2 + 1);
    uint32_t misfits = 0;
    depth_out = 1;
    if (nodes.empty()) throw std::runtime_error("fold_nodes: no nodes");
    for (size_t w = 0; w < source.size(); w++) {
        const GpuNode &b = nodes[source[w]];
        Entry e[4]; int ne = 0;
        auto take = [&](const float *lo, const float *hi, uint32_t child) {
            if (child == 0xFFFFFFFFu) return;                       // empty
            if (child & 0x80000000u) { e[ne++] = Entry{lo, hi, child}; return; } // leaf: stays
            if (child >= nodes.size()) throw std::runtime_error("fold_nodes: child index out of range");
            const GpuNode &c = nodes[child];                         // inner: its two children take its place
            const uint32_t cc[2] = {(uint32_t)c.child0, (uint32_t)c.child1};
            const float *clo[2] = {c.lo0, c.lo1}, *chi[2] = {c.hi0, c.hi1};
            for (int k = 0; k < 2; k++) if (cc[k] != 0xFFFFFFFFu) e[ne++] = Entry{clo[k], chi[k], cc[k]};
        };
        take(b.lo0, b.hi0, (uint32_t)b.child0);
        take(b.lo1, b.hi1, (uint32_t)b.child1);
        GpuNode4Q q;
        for (int k = 0; k < 4; k++) {
            if (k >= ne) { q.rec[k][0] = q.rec[k][1] = q.rec[k][2] = 0u; q.rec[k][3] = 0xFFFFFFFFu; continue; } // a point in the grid's border
            bool fits = true;
            for (int a = 0; a < 3; a++) q.rec[k][a] = grid_axis_word(e[k].lo[a], e[k].hi[a], grid.lo[a], grid.step[a], fits);
            if (!fits) misfits++;
            uint32_t child = e[k].child;
            if (!(child & 0x80000000u)) { // inner: gets the next wide node
                if (child >= nodes.size()) throw std::runtime_error("fold_nodes: child index out of range");
                source.push_back(child); level.push_back(level[w] + 1);
                if (level[w] + 1 > depth_out) depth_out = level[w] + 1;
                child = (uint32_t)(source.size() - 1);
                if (source.size() > nodes.size() + 1) throw std::runtime_error("fold_nodes: the two-box nodes do not form a tree");
            }
            q.rec[k][3] = child;
        }
        wide.push_back(q);
    }
    if (misfits) throw std::runtime_error("fold_nodes: " + std::to_string(misfits) + " node boxes do not fit the scene's grid");
}

} // namespace rtamd
