// Driver of the on-device scene-tree builder: device/rt_bvh_build.h has the kernels and the description of the algorithm.
#include "device/rt_bvh_build.h"
#include "device/rt_node_grid.h"
#include "host/device_build.h"
#include "host/fold_nodes.h"
#include "host/hip_check.h"
#include <cmath>
#include <stdexcept>
#include <string>
#include <vector>

namespace rtamd {

namespace {
__global__ void gather16_kernel(const float4 *in, float4 *out, const uint32_t *order, const uint8_t *last, uint32_t n, uint32_t quads, int mark_word, bool or_mark) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 *src = in + (size_t)order[i] * quads;
    float4 *dst = out + (size_t)i * quads;
    for (uint32_t q = 0; q < quads; q++) dst[q] = src[q];
    if (mark_word >= 0) {
        uint32_t *w = reinterpret_cast<uint32_t *>(dst) + mark_word;
        const uint32_t mark = last[i] ? 1u : 0u;
        *w = or_mark ? ((*w & ~1u) | mark) : mark; // or_mark: the word keeps its upper bits (a leaf mark of another tree in bit 0 is dropped)
    }
}
} // namespace

void join_root_box(const GpuNode *d_nodes, float lo[3], float hi[3]) {
    GpuNode root;
    HIP_CHECK(hipMemcpy(&root, d_nodes, sizeof root, hipMemcpyDeviceToHost));
    auto join = [&](const float *l, const float *h) { for (int k = 0; k < 3; k++) { if (l[k] < lo[k]) lo[k] = l[k]; if (h[k] > hi[k]) hi[k] = h[k]; } };
    if ((uint32_t)root.child0 != 0xFFFFFFFFu) join(root.lo0, root.hi0);
    if ((uint32_t)root.child1 != 0xFFFFFFFFu) join(root.lo1, root.hi1);
}

GpuNode4Q *widen_nodes(const GpuNode *d_nodes, uint32_t n, const NodeGrid &grid, uint32_t &n_out, uint32_t &depth_out) {
    std::vector<GpuNode> nodes(n ? n : 1);
    if (n) HIP_CHECK(hipMemcpy(nodes.data(), d_nodes, (size_t)n * sizeof(GpuNode), hipMemcpyDeviceToHost));
    else { GpuNode e{}; e.child0 = e.child1 = (int32_t)0xFFFFFFFFu; nodes[0] = e; }
    std::vector<GpuNode4Q> wide;
    fold_nodes(nodes, grid, wide, depth_out);
    n_out = (uint32_t)wide.size();
    GpuNode4Q *out = nullptr;
    HIP_CHECK(hipMalloc((void **)&out, wide.size() * sizeof(GpuNode4Q)));
    if (hipMemcpy(out, wide.data(), wide.size() * sizeof(GpuNode4Q), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(out); throw std::runtime_error("widen_nodes: upload failed"); }
    return out;
}

DeviceTree build_tree_on_device(const float *d_boxes, uint32_t n, float abs_pad, uint32_t max_depth) {
    DeviceTree t;
    if (n == 0) throw std::runtime_error("build_tree_on_device: no primitives");
    dev::BvbView B{};
    if (max_depth < 8 || max_depth > BVB_MAX_DEPTH) throw std::runtime_error("build_tree_on_device: max_depth out of range");
    B.boxes = d_boxes; B.n = n; B.abs_pad = abs_pad; B.max_depth = max_depth;
    const size_t max_open = (size_t)n / 2 + 2, max_nodes = 2 * (size_t)n + 2;
    std::vector<void *> temps;
    auto alloc = [&](size_t bytes, bool temp) { void *p = nullptr; HIP_CHECK(hipMalloc(&p, bytes)); if (temp) temps.push_back(p); return p; };
    try {
        B.prim_node = (uint32_t *)alloc((size_t)n * 4, true);
        B.nodes = (dev::BvbNode *)alloc(max_nodes * sizeof(dev::BvbNode), true);
        B.bins = (uint32_t *)alloc(max_open * BVB_NODE_BIN_WORDS * 4, true);
        B.dec = (dev::BvbDecision *)alloc(max_open * sizeof(dev::BvbDecision), true);
        B.open_cur = (uint32_t *)alloc(max_open * 4, true);
        B.open_next = (uint32_t *)alloc(max_open * 4, true);
        B.st = (dev::BvbState *)alloc(sizeof(dev::BvbState), true);
        t.order = B.order = (uint32_t *)alloc((size_t)n * 4, false);
        t.last = (uint8_t *)alloc((size_t)n, false);
        hipEvent_t e0, e1;
        HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
        hipStream_t stream = 0;
        HIP_CHECK(hipEventRecord(e0, stream));
        HIP_CHECK(hipMemsetAsync(B.nodes, 0, 2 * sizeof(dev::BvbNode), stream));
        HIP_CHECK(hipMemsetAsync(B.bins, 0, max_open * BVB_NODE_BIN_WORDS * 4, stream));
        HIP_CHECK(hipMemsetAsync(t.last, 0, n, stream));
        const uint32_t prim_blocks = (n + 255) / 256, prim_wgs = (n + BVB_THREADS - 1) / BVB_THREADS, open_blocks = (uint32_t)((max_open + 255) / 256);
        hipLaunchKernelGGL(dev::bvb_init_kernel, dim3(prim_blocks), dim3(256), 0, stream, B);
        hipLaunchKernelGGL(dev::bvb_root_kernel, dim3(1), dim3(64), 0, stream, B);
        for (int level = 0; level <= (int)max_depth; level++) {
            hipLaunchKernelGGL(dev::bvb_bin_kernel, dim3(prim_wgs), dim3(BVB_THREADS), 0, stream, B);
            hipLaunchKernelGGL(dev::bvb_split_kernel, dim3(open_blocks), dim3(256), 0, stream, B);
            hipLaunchKernelGGL(dev::bvb_number_kernel, dim3(1), dim3(BVB_THREADS), 0, stream, B);
            hipLaunchKernelGGL(dev::bvb_assign_kernel, dim3(prim_wgs), dim3(BVB_THREADS), 0, stream, B);
            std::swap(B.open_cur, B.open_next);
            hipLaunchKernelGGL(dev::bvb_next_level_kernel, dim3(1), dim3(64), 0, stream, B);
            if ((level & 7) == 7) { // every eighth level: is anything still open?
                dev::BvbState st;
                HIP_CHECK(hipMemcpyAsync(&st, B.st, sizeof st, hipMemcpyDeviceToHost, stream));
                HIP_CHECK(hipStreamSynchronize(stream));
                if (st.n_open == 0) break;
            }
        }
        hipLaunchKernelGGL(dev::bvb_layout_kernel, dim3(1), dim3(BVB_THREADS), 0, stream, B);
        dev::BvbState st;
        HIP_CHECK(hipMemcpyAsync(&st, B.st, sizeof st, hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        if (st.n_open != 0 || st.n_leaf_slots != n || st.n_nodes > max_nodes)
            throw std::runtime_error("build_tree_on_device: inconsistent tree (open " + std::to_string(st.n_open) + ", slots " + std::to_string(st.n_leaf_slots) + " of " + std::to_string(n) + ")");
        t.n_nodes = st.n_inner ? st.n_inner : 1u;
        t.depth = st.max_depth;
        t.nodes = B.out_nodes = (GpuNode *)alloc((size_t)t.n_nodes * sizeof(GpuNode), false);
        const uint32_t node_blocks = (st.n_nodes + 255) / 256;
        hipLaunchKernelGGL(dev::bvb_place_kernel, dim3(prim_blocks), dim3(256), 0, stream, B);
        hipLaunchKernelGGL(dev::bvb_emit_kernel, dim3(node_blocks), dim3(256), 0, stream, B);
        hipLaunchKernelGGL(dev::bvb_marks_kernel, dim3(node_blocks), dim3(256), 0, stream, B, t.last);
        HIP_CHECK(hipEventRecord(e1, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipEventElapsedTime(&t.build_ms, e0, e1));
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    } catch (...) {
        for (void *p : temps) (void)hipFree(p);
        free_device_tree(t);
        throw;
    }
    for (void *p : temps) (void)hipFree(p);
    return t;
}

void free_device_tree(DeviceTree &t) {
    if (t.nodes) (void)hipFree(t.nodes);
    if (t.order) (void)hipFree(t.order);
    if (t.last) (void)hipFree(t.last);
    t.nodes = nullptr; t.order = nullptr; t.last = nullptr;
}

void gather_records(const void *d_in, void *d_out, const DeviceTree &t, uint32_t n, uint32_t elem_bytes, int mark_word_offset, bool or_into_word) {
    if (elem_bytes % 16) throw std::runtime_error("gather_records: record size must be a multiple of 16");
    hipLaunchKernelGGL(gather16_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, (const float4 *)d_in, (float4 *)d_out, t.order, t.last, n, elem_bytes / 16, mark_word_offset, or_into_word);
    HIP_CHECK(hipGetLastError());
}

} // namespace rtamd
