// Host-side scene preparation: everything Scene::initBVH + Scene::initDistribution do in the
// reference (hw8/src/scene.cpp:65-78), re-expressed for the GPU:
//
//  1. figure order   — the reference's BVH build physically reorders `figures` with std::sort on the
//                      data3 vertex (hw8/src/include/bvh.h:60-109).  That order is part of the
//                      semantics (ties in t keep the lowest index; the light list is cut from it), so
//                      it is reproduced here by running the same libstdc++ algorithms with the same
//                      comparators on (key, index) pairs.
//  2. light order    — std::partition(emissive first) of a copy, then the same build over the first
//                      n (hw8/src/include/distributions.h:103-115).
//  3. GPU layouts    — the two trees are re-encoded as 64-byte two-box nodes over records stored in
//                      that order (rt_types.h); boxes are padded so the kernel's fast slab test is
//                      conservative, triangle records carry host-evaluated reference sub-expressions.
//
// Build with -ffp-contract=off (reference float semantics for the precomputed sub-expressions).
#include "scene_prep.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <limits>
#include <stdexcept>
#include <thread>

namespace rtamd {
namespace {

inline float smin(float a, float b) { return (b < a) ? b : a; } // std::min
inline float smax(float a, float b) { return (a < b) ? b : a; } // std::max

struct Box3 { float lo[3], hi[3]; };

inline float surface(const Box3 &b) { // primitives.cpp:158-161
    float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return 2 * (dx * dy + dx * dz + dy * dz);
}
inline void grow(Box3 &b, const Box3 &o) {
    for (int k = 0; k < 3; k++) { b.lo[k] = smin(b.lo[k], o.lo[k]); b.hi[k] = smax(b.hi[k], o.hi[k]); }
}

// Reference-topology builder over an index permutation.
struct RefNode { Box3 box; uint32_t left = 0, right = 0, first = 0, last = 0; };

class RefBuilder {
public:
    // keys[axis][tri] = data3 coordinate, boxes[tri] = triangle AABB; `order` is permuted in place.
    RefBuilder(const std::vector<float> *keys, const std::vector<Box3> &boxes, std::vector<uint32_t> &order)
        : keys_(keys), boxes_(boxes), order_(order), pairs_(order.size()), scores_(order.size()), suffix_(order.size()) {}

    std::vector<RefNode> nodes;
    uint32_t depth = 0;

    // The two subtrees of a node work on disjoint ranges of every array, so the large ones near the root are built on
    // separate threads into their own node lists and spliced in afterwards in the reference's (preorder) numbering.
    void run(uint32_t n) { nodes.clear(); depth = 0; depth = build(0, n, 1, nodes, 0); }

private:
    struct KI { float k; uint32_t i; };
    const std::vector<float> *keys_;
    const std::vector<Box3> &boxes_;
    std::vector<uint32_t> &order_;
    std::vector<KI> pairs_;
    std::vector<float> scores_;
    std::vector<float> suffix_;

    void sort_axis(uint32_t first, uint32_t last, int axis) { // bvh.h:60-65
        const std::vector<float> &key = keys_[axis];
        for (uint32_t i = first; i < last; i++) pairs_[i] = KI{key[order_[i]], order_[i]};
        std::sort(pairs_.begin() + first, pairs_.begin() + last, [](const KI &l, const KI &r) { return l.k < r.k; });
        for (uint32_t i = first; i < last; i++) order_[i] = pairs_[i].i;
    }
    std::pair<float, uint32_t> best_split(uint32_t first, uint32_t last) { // bvh.h:34-54
        uint32_t n = last - first;
        float *sc = scores_.data() + first;
        sc[0] = 0;
        Box3 pre = boxes_[order_[first]];
        for (size_t i = 1; i < n; i++) {
            sc[i] = surface(pre) * i;
            grow(pre, boxes_[order_[first + i]]);
        }
        Box3 suf = boxes_[order_[last - 1]];
        for (size_t i = n - 1; i >= 1; i--) {
            sc[i] += surface(suf) * (n - i);
            grow(suf, boxes_[order_[first + i - 1]]);
        }
        std::pair<float, uint32_t> ans = {sc[1], first + 1};
        for (size_t i = 2; i < n; i++)
            if (sc[i] < ans.first) ans = {sc[i], (uint32_t)(first + i)};
        return ans;
    }
    // Builds the subtree over [first,last) into `out` (indices relative to `out`); returns its depth.
    uint32_t build(uint32_t first, uint32_t last, uint32_t d, std::vector<RefNode> &out, int fork_level) { // bvh.h:67-109
        uint32_t deepest = d;
        RefNode cur;
        cur.first = first; cur.last = last;
        if (first < last) {
            cur.box = boxes_[order_[first]];
            for (uint32_t i = first + 1; i < last; i++) grow(cur.box, boxes_[order_[i]]);
        } else memset(&cur.box, 0, sizeof cur.box);
        uint32_t pos = (uint32_t)out.size();
        out.push_back(cur);
        if (last - first <= 1) return deepest;
        sort_axis(first, last, 0); auto sx = best_split(first, last);
        sort_axis(first, last, 1); auto sy = best_split(first, last);
        sort_axis(first, last, 2); auto sz = best_split(first, last);
        float best = smin(sx.first, smin(sy.first, sz.first));
        if (best >= surface(cur.box) * (last - first)) return deepest; // leaf with >1 triangles, z-sorted
        uint32_t mid;
        if (best == sx.first) { mid = sx.second; sort_axis(first, last, 0); }
        else if (best == sy.first) { mid = sy.second; sort_axis(first, last, 1); }
        else { mid = sz.second; sort_axis(first, last, 2); }
        if (fork_level < 3 && last - first > 16384) { // right subtree on another thread, spliced behind the left one
            std::vector<RefNode> right_nodes;
            uint32_t right_depth = 0;
            std::thread worker([&] { right_depth = build(mid, last, d + 1, right_nodes, fork_level + 1); });
            out[pos].left = (uint32_t)out.size();
            uint32_t left_depth = build(first, mid, d + 1, out, fork_level + 1);
            worker.join();
            uint32_t shift = (uint32_t)out.size();
            out[pos].right = shift;
            for (RefNode n : right_nodes) {
                if (n.left != 0) { n.left += shift; n.right += shift; }
                out.push_back(n);
            }
            return left_depth > right_depth ? left_depth : right_depth;
        }
        uint32_t l = (uint32_t)out.size();
        out[pos].left = l;
        uint32_t dl = build(first, mid, d + 1, out, fork_level + 1);
        uint32_t r = (uint32_t)out.size();
        out[pos].right = r;
        uint32_t dr = build(mid, last, d + 1, out, fork_level + 1);
        return dl > dr ? dl : dr;
    }
};

// Conservative padding for the kernels' reciprocal-multiply slab test: 2^-17 relative to the coordinate magnitude (64 ulp) on every
// face, plus abs_pad = 2^-18 x the largest |coordinate| of scene and camera.  The absolute part is what makes the walk independent of
// the tree: a triangle test accepts points up to a few ulp OF THE LARGEST COORDINATES INVOLVED outside the true triangle (p = o + t d),
// so a hit on an edge that lies in a box face at a small coordinate (x = 0: relative padding vanishes) must not be pruned by one
// tree's boxes and kept by another's — the exactness gate (device/rt_exact.h) relies on every such hit, and its runner-up, being seen.
inline void pad_box(const Box3 &b, float lo[3], float hi[3], float abs_pad) {
    for (int k = 0; k < 3; k++) {
        float mag = smax(std::fabs(b.lo[k]), std::fabs(b.hi[k]));
        float pad = mag * 7.62939453125e-06f + abs_pad + 1e-30f;
        lo[k] = b.lo[k] - pad;
        hi[k] = b.hi[k] + pad;
    }
}
inline float scene_abs_pad(const std::vector<Box3> &boxes, const rt_camera &cam) {
    float m = 0.f;
    for (int k = 0; k < 3; k++) m = smax(m, std::fabs(cam.position[k]));
    for (const Box3 &b : boxes) for (int k = 0; k < 3; k++) m = smax(m, smax(std::fabs(b.lo[k]), std::fabs(b.hi[k])));
    float scale = 1.f;
    if (const char *e = getenv("RTAMD_BOX_PAD_SCALE")) scale = (float)atof(e); // experiments
    return m * 3.814697265625e-06f * scale; // 2^-18
}

// Re-encode a reference tree as two-box GpuNodes (child reference: inner node index, or
// 0x80000000|first for a leaf, 0xFFFFFFFF for an empty leaf).  `leaf_last` receives the index of
// the last primitive of every leaf; the kernels walk a leaf until they meet that mark.
void encode_tree(const std::vector<RefNode> &ref, std::vector<GpuNode> &out, std::vector<uint32_t> &leaf_last, float abs_pad = 0.f) {
    out.clear();
    leaf_last.clear();
    auto empty_child = [](float lo[3], float hi[3], int32_t &child, int32_t &cnt) {
        for (int k = 0; k < 3; k++) { lo[k] = 3.0e38f; hi[k] = 3.0e38f; }
        child = (int32_t)0xFFFFFFFFu; cnt = 0;
    };
    auto leaf_ref = [&](const RefNode &l, int32_t &child, int32_t &cnt) {
        child = (int32_t)(0x80000000u | l.first); cnt = (int32_t)(l.last - l.first);
        leaf_last.push_back(l.last - 1);
    };
    if (ref.empty() || ref[0].last == ref[0].first) { // no primitives: one node, two empty leaves
        GpuNode g;
        empty_child(g.lo0, g.hi0, g.child0, g.cnt0);
        empty_child(g.lo1, g.hi1, g.child1, g.cnt1);
        out.push_back(g);
        return;
    }
    if (ref[0].left == 0) { // root is a leaf: wrap it
        GpuNode g;
        pad_box(ref[0].box, g.lo0, g.hi0, abs_pad);
        leaf_ref(ref[0], g.child0, g.cnt0);
        empty_child(g.lo1, g.hi1, g.child1, g.cnt1);
        out.push_back(g);
        return;
    }
    // Inner reference nodes get GPU indices in DFS preorder (root = 0).
    std::vector<int32_t> gpu_index(ref.size(), -1);
    int32_t next = 0;
    for (size_t i = 0; i < ref.size(); i++)
        if (ref[i].left != 0) gpu_index[i] = next++;
    out.resize(next);
    for (size_t i = 0; i < ref.size(); i++) {
        if (ref[i].left == 0) continue;
        GpuNode &g = out[gpu_index[i]];
        const RefNode &l = ref[ref[i].left], &r = ref[ref[i].right];
        pad_box(l.box, g.lo0, g.hi0, abs_pad);
        pad_box(r.box, g.lo1, g.hi1, abs_pad);
        if (l.left == 0) leaf_ref(l, g.child0, g.cnt0);
        else { g.child0 = gpu_index[ref[i].left]; g.cnt0 = 0; }
        if (r.left == 0) leaf_ref(r, g.child1, g.cnt1);
        else { g.child1 = gpu_index[ref[i].right]; g.cnt1 = 0; }
    }
}

// The reference tree as the exact walks read it (unpadded boxes, preorder numbering of the builder).
void encode_ref_tree(const std::vector<RefNode> &ref, std::vector<GpuRefNode> &out) {
    out.resize(ref.empty() ? 1 : ref.size());
    if (ref.empty()) { memset(&out[0], 0, sizeof out[0]); return; }
    for (size_t i = 0; i < ref.size(); i++) {
        GpuRefNode &g = out[i];
        for (int k = 0; k < 3; k++) { g.mn[k] = ref[i].box.lo[k]; g.mx[k] = ref[i].box.hi[k]; }
        g.left = ref[i].left; g.right = ref[i].right; g.first = ref[i].first; g.last = ref[i].last;
        g.pad0 = g.pad1 = 0;
    }
}

// Renumber the inner nodes so that the first `top` of them are the top of the tree in breadth-first order (root = 0 stays):
// the persistent kernel's LDS treelet experiment (device/rt_persistent.h, PT_TREELET) keeps nodes [0, top) in LDS.  Any numbering
// is valid for every kernel (children are explicit references); the rest keep their depth-first order.
void bfs_top_first(std::vector<GpuNode> &nodes, uint32_t top) {
    const uint32_t n = (uint32_t)nodes.size();
    if (n <= 1 || top <= 1) return;
    std::vector<uint32_t> order; // new index -> old index
    order.reserve(n);
    std::vector<uint8_t> taken(n, 0);
    order.push_back(0); taken[0] = 1;
    for (size_t head = 0; head < order.size() && order.size() < top; head++) {
        const GpuNode &g = nodes[order[head]];
        for (int32_t c : {g.child0, g.child1})
            if (c >= 0 && (uint32_t)c < n && !taken[c] && order.size() < top) { order.push_back((uint32_t)c); taken[c] = 1; }
    }
    for (uint32_t i = 0; i < n; i++) if (!taken[i]) order.push_back(i);
    std::vector<uint32_t> new_of(n);
    for (uint32_t i = 0; i < n; i++) new_of[order[i]] = i;
    std::vector<GpuNode> out(n);
    for (uint32_t i = 0; i < n; i++) {
        GpuNode g = nodes[order[i]];
        if (g.child0 >= 0) g.child0 = (int32_t)new_of[g.child0];
        if (g.child1 >= 0) g.child1 = (int32_t)new_of[g.child1];
        out[i] = g;
    }
    nodes.swap(out);
}

struct V3 { float x, y, z; };
inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 crossr(V3 a, V3 o) { return {a.z * o.y - a.y * o.z, a.x * o.z - a.z * o.x, a.y * o.x - a.x * o.y}; } // vec3.h:57-59

const float magic1[] = {0.239, 0.419, 0.533};       // primitives.cpp:82
const float magic2[] = {0.35743, 0.66682, 0.69695}; // primitives.cpp:83

TriIsect make_isect(const float *p /* data, data2, data3 */) {
    V3 d0{p[0], p[1], p[2]}, d1{p[3], p[4], p[5]}, a{p[6], p[7], p[8]};
    V3 b = sub(d0, a), c = sub(d1, a);
    V3 n = crossr(b, c);
    TriIsect t;
    t.ax = a.x; t.ay = a.y; t.az = a.z;
    t.nx = n.x; t.ny = n.y; t.nz = n.z;
    t.a1 = magic1[0] * b.x + magic1[1] * b.y + magic1[2] * b.z;
    t.b1 = magic1[0] * c.x + magic1[1] * c.y + magic1[2] * c.z;
    t.a2 = magic2[0] * b.x + magic2[1] * b.y + magic2[2] * b.z;
    t.b2 = magic2[0] * c.x + magic2[1] * c.y + magic2[2] * c.z;
    t.den = t.b1 * t.a2 - t.a1 * t.b2;
    t.pad = 0;
    return t;
}

} // namespace

// Separation depths of neighbouring lights in a reference light tree + sparse table for range minima (see scene_prep.h).
static void build_light_sep(const std::vector<RefNode> &rn, uint32_t nl, std::vector<uint16_t> &table, uint32_t &table_levels) {
    std::vector<uint16_t> sep(nl ? nl : 1, 0);
    if (nl > 1) {
        std::vector<std::pair<uint32_t, uint32_t>> todo; // node, depth
        todo.push_back({0u, 0u});
        while (!todo.empty()) {
            auto [node, depth] = todo.back();
            todo.pop_back();
            const RefNode &nd = rn[node];
            if (nd.left == 0) { // leaf: its lights are added left to right, so the LAST boundary is the shallowest
                for (uint32_t b = nd.first; b + 1 < nd.last; b++) sep[b] = (uint16_t)std::min<uint32_t>(65535u, 32768u + (nd.last - 2 - b));
                continue;
            }
            sep[rn[nd.right].first - 1] = (uint16_t)depth; // boundary between the last light of the left child and the first of the right
            todo.push_back({nd.left, depth + 1});
            todo.push_back({nd.right, depth + 1});
        }
    }
    uint32_t levels = 1;
    while ((1u << levels) < (nl ? nl : 1)) levels++;
    table_levels = levels;
    table.assign((size_t)levels * (nl ? nl : 1), 0);
    const size_t stride = nl ? nl : 1;
    for (size_t b = 0; b < stride; b++) table[b] = sep[b];
    for (uint32_t j = 1; j < levels; j++)
        for (size_t b = 0; b < stride; b++) {
            size_t o2 = b + (1u << (j - 1));
            uint16_t a = table[(j - 1) * stride + b], c = o2 < stride ? table[(j - 1) * stride + o2] : a;
            table[j * stride + b] = a < c ? a : c;
        }
}

void prepare_scene(const rt_scene_desc &d, PreparedScene &out, bool tree_on_device) {
    const uint32_t n = d.n_triangles;
    if (n >= 0x00FFFFFFu) throw std::runtime_error("too many triangles (limit 2^24-2: the hit word keeps 24 bits of figure index, device/rt_exact.h)");
    if (n && (!d.positions || !d.material_index)) throw std::runtime_error("scene has triangles but no positions/material_index");
    for (uint32_t i = 0; i < n; i++)
        if (d.material_index[i] >= d.n_materials) throw std::runtime_error("triangle material index out of range");

    // ---- 1. figure order (scene BVH) --------------------------------------------------------------
    std::vector<float> keys[3];
    std::vector<Box3> boxes(n);
    for (int k = 0; k < 3; k++) keys[k].resize(n);
    for (uint32_t i = 0; i < n; i++) {
        const float *p = d.positions + 9 * (size_t)i;
        for (int k = 0; k < 3; k++) {
            keys[k][i] = p[6 + k]; // data3
            boxes[i].lo[k] = smin(p[6 + k], smin(p[k], p[3 + k])); // primitives.cpp:130-141
            boxes[i].hi[k] = smax(p[6 + k], smax(p[k], p[3 + k]));
        }
    }
    out.figure_order.resize(n);
    for (uint32_t i = 0; i < n; i++) out.figure_order[i] = i;
    std::vector<uint32_t> scene_leaf_last, light_leaf_last;
    out.box_pad = scene_abs_pad(boxes, d.camera);
    if (tree_on_device) {
        // RT_BUILD_DEVICE_BVH: no replay of the reference's builder -- the figure order is the LOAD order, the records below stay in
        // it (tri_box doubles as the builder's input) and rtamd_build.hip makes the tree and the leaf order on the GPU
        out.nodes.clear();
        encode_ref_tree(std::vector<RefNode>(), out.ref_nodes);
    } else {
        RefBuilder scene_builder(keys, boxes, out.figure_order);
        scene_builder.run(n);
        out.bvh_depth = scene_builder.depth;
        out.n_ref_nodes = (uint32_t)scene_builder.nodes.size();
        encode_tree(scene_builder.nodes, out.nodes, scene_leaf_last, out.box_pad);
        bfs_top_first(out.nodes, 512);
        // Walk boxes: the box of the figure's reference leaf (a few leaves hold several triangles): whatever the reference can reach
        // through its leaf box is then inside a box of ours.  (Widening each box by how far outside the triangle the reference's test
        // can still report a hit -- delta cot(phi), rt_exact.h -- was tried: the few triangles whose plane contains the projection's
        // kernel get scene-sized boxes, their test accepts a sliver of every ray, and 12 % of the queries end in the exact walk.)
        out.walk_box.assign((size_t)(n ? n : 1) * 8, 0.f);
        for (const RefNode &rn : scene_builder.nodes)
            if (rn.left == 0)
                for (uint32_t i = rn.first; i < rn.last; i++)
                    for (int k = 0; k < 3; k++) { out.walk_box[8 * (size_t)i + k] = rn.box.lo[k]; out.walk_box[8 * (size_t)i + 4 + k] = rn.box.hi[k]; }
        encode_ref_tree(scene_builder.nodes, out.ref_nodes);
    }

    // ---- 2. light order ---------------------------------------------------------------------------
    auto emissive = [&](uint32_t tri) { // distributions.h:104-109: the FACTOR decides, not the texture
        const rt_material &m = d.materials[d.material_index[tri]];
        return !(m.emission[0] == 0 && m.emission[1] == 0 && m.emission[2] == 0);
    };
    std::vector<uint32_t> lorder = out.figure_order; // the copy FiguresMix receives by value
    uint32_t n_lights = (uint32_t)(std::partition(lorder.begin(), lorder.end(), emissive) - lorder.begin());
    RefBuilder light_builder(keys, boxes, lorder);
    light_builder.run(n_lights);
    out.light_bvh_depth = light_builder.depth;
    encode_tree(light_builder.nodes, out.light_nodes, light_leaf_last, out.box_pad);
    encode_ref_tree(light_builder.nodes, out.ref_light_nodes);
    build_light_sep(light_builder.nodes, n_lights, out.light_sep, out.light_sep_levels);
    out.light_walk_box.assign((size_t)(n_lights ? n_lights : 1) * 8, 0.f);
    for (const RefNode &rn : light_builder.nodes)
        if (rn.left == 0)
            for (uint32_t i = rn.first; i < rn.last; i++)
                for (int k = 0; k < 3; k++) { out.light_walk_box[8 * (size_t)i + k] = rn.box.lo[k]; out.light_walk_box[8 * (size_t)i + 4 + k] = rn.box.hi[k]; }
    out.light_order.assign(lorder.begin(), lorder.begin() + n_lights);

    // ---- 3. records ---------------------------------------------------------------------------------
    out.isect.resize(n);
    out.shade.resize(n);
    out.tri_box.assign((size_t)(n ? n : 1) * 8, 0.f);
    float max_coord = 0.f;
    for (int k = 0; k < 3; k++) max_coord = smax(max_coord, std::fabs(d.camera.position[k]));
    for (uint32_t i = 0; i < n; i++) {
        uint32_t src = out.figure_order[i];
        for (int k = 0; k < 3; k++) {
            out.tri_box[8 * (size_t)i + k] = boxes[src].lo[k];
            out.tri_box[8 * (size_t)i + 4 + k] = boxes[src].hi[k];
            max_coord = smax(max_coord, smax(std::fabs(boxes[src].lo[k]), std::fabs(boxes[src].hi[k])));
        }
        out.isect[i] = make_isect(d.positions + 9 * (size_t)src);
        out.isect[i].pad = i << 1; // figure index; bit 0 (last of its leaf) is set below / by the GPU builder's gather
        TriShade &s = out.shade[i];
        memset(&s, 0, sizeof s);
        if (d.normals) {
            const float *q = d.normals + 9 * (size_t)src;
            for (int k = 0; k < 3; k++) { s.n3[k] = q[6 + k]; s.dn1[k] = q[k] - q[6 + k]; s.dn2[k] = q[3 + k] - q[6 + k]; }
        }
        if (d.tangents) {
            const float *q = d.tangents + 12 * (size_t)src;
            for (int k = 0; k < 3; k++) { s.t3[k] = q[8 + k]; s.dt1[k] = q[k] - q[8 + k]; s.dt2[k] = q[4 + k] - q[8 + k]; }
            s.tanw = q[3];
        }
        if (d.texcoords) {
            const float *q = d.texcoords + 6 * (size_t)src;
            for (int k = 0; k < 2; k++) { s.uv3[k] = q[4 + k]; s.duv1[k] = q[k] - q[4 + k]; s.duv2[k] = q[2 + k] - q[4 + k]; }
        }
        s.material = d.material_index[src];
        s.orig = src;
    }
    out.box_c2 = max_coord * 9.5367431640625e-07f; // 2^-20
    for (uint32_t i : scene_leaf_last) out.isect[i].pad |= 1u;
    out.tripwires.clear(); out.n_tripwire_groups = 0;
    if (!out.walk_box.empty()) { // replay mode: the tripwires of scene_prep.h
        // |den| / (|n| |k|) = |cos| of the angle between the triangle's normal and the projection's kernel k = magic1 x magic2: how far
        // outside the triangle the test can still accept a point grows like (rounding of the hit point) / cos.  Below 2^-14 (20 triangles of the
        // benchmark scene; 8 lie below 1e-5, one at 4e-8: its test accepted a point 16 units away) the walkers' look-behind and the gate's
        // window (rt_exact.h) no longer cover it; every ray pays for the test, so the list is kept short.
        const double m1[3] = {0.239f, 0.419f, 0.533f}, m2[3] = {0.35743f, 0.66682f, 0.69695f};
        const double kx = m1[1] * m2[2] - m1[2] * m2[1], ky = m1[2] * m2[0] - m1[0] * m2[2], kz = m1[0] * m2[1] - m1[1] * m2[0];
        const double kn = std::sqrt(kx * kx + ky * ky + kz * kz);
        const double tripwire_cos = getenv("RTAMD_TRIPWIRE_COS") ? atof(getenv("RTAMD_TRIPWIRE_COS")) : 6.103515625e-05; // 2^-14
        struct Wire { uint64_t key; float lo[3], hi[3]; uint32_t figure; };
        std::vector<Wire> wires;
        float slo[3] = {3e38f, 3e38f, 3e38f}, shi[3] = {-3e38f, -3e38f, -3e38f};
        for (uint32_t i = 0; i < n; i++) for (int k = 0; k < 3; k++) { slo[k] = smin(slo[k], out.walk_box[8 * (size_t)i + k]); shi[k] = smax(shi[k], out.walk_box[8 * (size_t)i + 4 + k]); }
        for (uint32_t i = 0; i < n; i++) {
            const TriIsect &T = out.isect[i];
            const double nn = std::sqrt((double)T.nx * T.nx + (double)T.ny * T.ny + (double)T.nz * T.nz);
            if (!(nn > 0.0)) continue;                                       // no plane: its test never passes (t is NaN or infinite)
            if (std::fabs((double)T.den) >= tripwire_cos * nn * kn) continue;
            Wire w;
            const float pad = 1e-4f * max_coord + 1e-30f;                    // far more than any rounding of the reference's box test
            uint64_t key = 0;
            for (int k = 0; k < 3; k++) {
                w.lo[k] = out.walk_box[8 * (size_t)i + k] - pad; w.hi[k] = out.walk_box[8 * (size_t)i + 4 + k] + pad;
                const double c = 0.5 * ((double)w.lo[k] + w.hi[k]), e = (double)shi[k] - slo[k];
                const uint64_t q = e > 0 ? (uint64_t)std::fmin(1023.0, std::fmax(0.0, (c - slo[k]) / e * 1024.0)) : 0;
                for (int b = 0; b < 10; b++) key |= ((q >> b) & 1ull) << (3 * b + k);       // Morton code: neighbours end up in one group
            }
            w.key = key;
            w.figure = i;
            wires.push_back(w);
        }
        std::sort(wires.begin(), wires.end(), [](const Wire &a, const Wire &b) { return a.key < b.key; });
        // Groups: the wires lie in a few far-apart families (the band of a sphere's triangles whose normals are perpendicular to k), and a group's
        // box must stay small — a ray that pierces it pays for its members.  Neighbouring runs of the Morton order are merged, cheapest union
        // (by surface area) first, until at most four are left.
        struct Run { size_t first, last; float lo[3], hi[3]; };
        std::vector<Run> runs;
        for (size_t m = 0; m < wires.size(); m++) { Run r; r.first = m; r.last = m + 1; for (int k = 0; k < 3; k++) { r.lo[k] = wires[m].lo[k]; r.hi[k] = wires[m].hi[k]; } runs.push_back(r); }
        auto area_of_union = [](const Run &a, const Run &b) {
            double e[3];
            for (int k = 0; k < 3; k++) e[k] = (double)smax(a.hi[k], b.hi[k]) - (double)smin(a.lo[k], b.lo[k]);
            return e[0] * e[1] + e[1] * e[2] + e[2] * e[0];
        };
        while (runs.size() > 4) {
            size_t best = 0; double best_area = 1e300;
            for (size_t j = 0; j + 1 < runs.size(); j++) { const double ar = area_of_union(runs[j], runs[j + 1]); if (ar < best_area) { best_area = ar; best = j; } }
            Run &a = runs[best]; const Run &b2 = runs[best + 1];
            a.last = b2.last;
            for (int k = 0; k < 3; k++) { a.lo[k] = smin(a.lo[k], b2.lo[k]); a.hi[k] = smax(a.hi[k], b2.hi[k]); }
            runs.erase(runs.begin() + (long)best + 1);
        }
        const size_t n_groups = runs.size();
        out.n_tripwire_groups = (uint32_t)n_groups;
        if (getenv("RTAMD_DUMP_TRIPWIRES")) // diagnostic
            for (const Run &r : runs) fprintf(stderr, "[rtamd] tripwire group: %zu wires, box %.3f %.3f %.3f .. %.3f %.3f %.3f\n", r.last - r.first, r.lo[0], r.lo[1], r.lo[2], r.hi[0], r.hi[1], r.hi[2]);
        out.tripwires.assign((n_groups + wires.size()) * 8, 0.f);
        for (size_t g = 0; g < n_groups; g++) {
            const size_t first = runs[g].first, last = runs[g].last;
            float *rec = &out.tripwires[8 * g];
            for (int k = 0; k < 3; k++) { rec[k] = runs[g].lo[k]; rec[4 + k] = runs[g].hi[k]; }
            for (size_t m = first; m < last; m++) {
                float *mr = &out.tripwires[8 * (n_groups + m)];
                for (int k = 0; k < 3; k++) { mr[k] = wires[m].lo[k]; mr[4 + k] = wires[m].hi[k]; }
                memcpy(&mr[3], &wires[m].figure, 4);
            }
            const uint32_t first_rec = (uint32_t)(n_groups + first), count = (uint32_t)(last - first);
            memcpy(&rec[3], &first_rec, 4); memcpy(&rec[7], &count, 4);
        }
    }
    out.lights.resize(n_lights);
    for (uint32_t i = 0; i < n_lights; i++) {
        uint32_t src = out.light_order[i];
        const float *p = d.positions + 9 * (size_t)src;
        LightRec &L = out.lights[i];
        memset(&L, 0, sizeof L);
        L.isect = make_isect(p);
        for (int k = 0; k < 3; k++) { L.b[k] = p[k] - p[6 + k]; L.c[k] = p[3 + k] - p[6 + k]; }
        float nl2 = L.isect.nx * L.isect.nx + L.isect.ny * L.isect.ny + L.isect.nz * L.isect.nz;
        float nl = (float)std::sqrt((double)nl2);
        L.point_prob = (float)(1.0 / (0.5 * (double)nl)); // distributions.h:78
        { // the box the kernel used to form per hit: a, a + b, a + c in float (not the original vertices)
            const float av[3] = {L.isect.ax, L.isect.ay, L.isect.az};
            for (int k = 0; k < 3; k++) {
                const float pb = av[k] + L.b[k], pc = av[k] + L.c[k];
                L.box_lo[k] = std::fmin(av[k], std::fmin(pb, pc));
                L.box_hi[k] = std::fmax(av[k], std::fmax(pb, pc));
            }
        }
        if (d.normals) {
            const float *q = d.normals + 9 * (size_t)src;
            for (int k = 0; k < 3; k++) { L.n3[k] = q[6 + k]; L.dn1[k] = q[k] - q[6 + k]; L.dn2[k] = q[3 + k] - q[6 + k]; }
        }
    }

    for (uint32_t i : light_leaf_last) out.lights[i].isect.pad = 1;

    // ---- materials, images ------------------------------------------------------------------------
    out.images.clear();
    out.texels.clear();
    auto add_image = [&](const rt_image &im) -> int32_t {
        if (im.width <= 0 || im.height <= 0 || !im.rgb) throw std::runtime_error("image with no pixels");
        GpuImage g;
        g.offset = out.texels.size();
        g.width = im.width; g.height = im.height;
        size_t bytes = (size_t)im.width * im.height * 3;
        if (bytes >= 0xFFFFFFF0ull) throw std::runtime_error("image too large (the kernels address its texels with 32 bits)");
        out.texels.insert(out.texels.end(), im.rgb, im.rgb + bytes);
        out.texels.push_back(0); // a texel is fetched with one 4-byte load (rt_device.h load_texel): one spare byte behind the last one
        while (out.texels.size() % 16) out.texels.push_back(0);
        out.images.push_back(g);
        return (int32_t)out.images.size() - 1;
    };
    for (uint32_t i = 0; i < d.n_images; i++) add_image(d.images[i]);
    out.env_image = d.environment_map ? add_image(*d.environment_map) : -1;
    auto slot = [&](int32_t tex) -> int32_t {
        if (tex < 0) return -1;
        if ((uint32_t)tex >= d.n_textures) throw std::runtime_error("material texture index out of range");
        uint32_t src = d.texture_source[tex];
        if (src >= d.n_images) throw std::runtime_error("texture source out of range");
        return (int32_t)src;
    };
    out.materials.resize(d.n_materials);
    for (uint32_t i = 0; i < d.n_materials; i++) {
        const rt_material &m = d.materials[i];
        GpuMaterial &g = out.materials[i];
        for (int k = 0; k < 3; k++) { g.base_color[k] = m.base_color[k]; g.emission[k] = m.emission[k]; }
        g.metallic_factor = m.metallic_factor; g.roughness_factor = m.roughness_factor;
        g.base_color_tex = slot(m.base_color_texture); g.emissive_tex = slot(m.emissive_texture);
        g.metallic_roughness_tex = slot(m.metallic_roughness_texture); g.normal_tex = slot(m.normal_texture);
    }
    // sRGB decode table: loadSingleFromTexture (scene.cpp:9-16) applies std::pow(float, 2.2f) to
    // float(1./255)*byte; only 256 inputs exist, so the host libm (the one the reference itself
    // would call) evaluates them once and the kernel looks them up: bit-exact and cheaper.
    for (int b = 0; b < 256; b++) out.srgb_lut[b] = std::pow((float)(1. / 255) * (1.f * (float)b), 2.2f);
}


void prepare_scene_hw6(const rt_scene_desc &d, PreparedScene6 &out, bool tree_on_device) {
    const uint32_t n = d.n_triangles;
    if (n >= 0x7FFFFFFFu) throw std::runtime_error("too many triangles (limit 2^31-2)");
    if (n && (!d.positions || !d.material_index)) throw std::runtime_error("scene has triangles but no positions/material_index");
    for (uint32_t i = 0; i < n; i++)
        if (d.material_index[i] >= d.n_materials) throw std::runtime_error("triangle material index out of range");
    std::vector<float> zero_keys[3], keys[3];
    std::vector<Box3> boxes(n);
    for (int k = 0; k < 3; k++) { zero_keys[k].assign(n, 0.f); keys[k].resize(n); }
    for (uint32_t i = 0; i < n; i++) {
        const float *p = d.positions + 9 * (size_t)i;
        for (int k = 0; k < 3; k++) {
            keys[k][i] = p[6 + k];
            boxes[i].lo[k] = smin(p[6 + k], smin(p[k], p[3 + k])); // hw6/src/primitives.cpp:173-183
            boxes[i].hi[k] = smax(p[6 + k], smax(p[k], p[3 + k]));
        }
    }
    // 1. the reference's figure order: hw6's BVH sorts on Figure::position == (0,0,0) for every triangle
    //    (hw6/src/include/bvh.h:61-63); the resulting permutation is what introsort does with all-equal keys.
    out.figure_order.resize(n);
    for (uint32_t i = 0; i < n; i++) out.figure_order[i] = i;
    RefBuilder ref(zero_keys, boxes, out.figure_order);
    ref.run(n);
    out.ref_bvh_depth = ref.depth;
    encode_ref_tree(ref.nodes, out.ref_nodes);
    std::vector<uint32_t> ref_pos(n);
    for (uint32_t i = 0; i < n; i++) ref_pos[out.figure_order[i]] = i;
    // the reference leaf box of every triangle (LOAD order): what the walkers' own tree is built over (see PreparedScene::walk_box)
    std::vector<Box3> leaf_box = boxes;
    for (const RefNode &rn : ref.nodes)
        if (rn.left == 0)
            for (uint32_t i = rn.first; i < rn.last; i++) leaf_box[out.figure_order[i]] = rn.box;
    // 2. light order + light tree (reference topology: it fixes the order of the float additions)
    auto emissive = [&](uint32_t tri) {
        const rt_material &m = d.materials[d.material_index[tri]];
        return !(m.emission[0] == 0 && m.emission[1] == 0 && m.emission[2] == 0);
    };
    std::vector<uint32_t> lorder = out.figure_order;
    uint32_t n_lights = (uint32_t)(std::partition(lorder.begin(), lorder.end(), emissive) - lorder.begin());
    RefBuilder light_builder(zero_keys, boxes, lorder);
    light_builder.run(n_lights);
    out.light_bvh_depth = light_builder.depth;
    std::vector<uint32_t> light_leaf_last, scene_leaf_last;
    out.box_pad = scene_abs_pad(boxes, d.camera);
    encode_tree(light_builder.nodes, out.light_nodes, light_leaf_last, out.box_pad);
    encode_ref_tree(light_builder.nodes, out.ref_light_nodes);
    out.light_order.assign(lorder.begin(), lorder.begin() + n_lights);
    // 3. own scene tree: the same full-sweep SAH builder keyed on the data3 vertex (as hw8 does) -- or, when the caller builds the
    //    tree on the GPU (device/rt_bvh_build.h), the records stay in LOAD order and the boxes go along
    std::vector<uint32_t> my_order(n);
    for (uint32_t i = 0; i < n; i++) my_order[i] = i;
    if (tree_on_device) {
        out.boxes8.assign((size_t)n * 8, 0.f);
        for (uint32_t i = 0; i < n; i++)
            for (int k = 0; k < 3; k++) { out.boxes8[8 * (size_t)i + k] = leaf_box[i].lo[k]; out.boxes8[8 * (size_t)i + 4 + k] = leaf_box[i].hi[k]; }
    } else {
        RefBuilder mine(keys, boxes, my_order);
        mine.run(n);
        out.bvh_depth = mine.depth;
        encode_tree(mine.nodes, out.nodes, scene_leaf_last, out.box_pad);
    }
    auto make = [&](uint32_t src) {
        const float *p = d.positions + 9 * (size_t)src;
        Tri6 t;
        memset(&t, 0, sizeof t);
        V3 a{p[6], p[7], p[8]}, b = sub(V3{p[0], p[1], p[2]}, a), c = sub(V3{p[3], p[4], p[5]}, a);
        V3 nn = crossr(b, c);
        t.a[0] = a.x; t.a[1] = a.y; t.a[2] = a.z; t.b[0] = b.x; t.b[1] = b.y; t.b[2] = b.z;
        t.c[0] = c.x; t.c[1] = c.y; t.c[2] = c.z; t.n[0] = nn.x; t.n[1] = nn.y; t.n[2] = nn.z;
        t.material = d.material_index[src];
        float nl = (float)std::sqrt((double)(nn.x * nn.x + nn.y * nn.y + nn.z * nn.z));
        t.point_prob = (float)(1.0 / (0.5 * (double)nl)); // hw6/src/include/distributions.h:126
        return t;
    };
    out.tris.resize(n);
    for (uint32_t i = 0; i < n; i++) { out.tris[i] = make(my_order[i]); out.tris[i].ref_index = ref_pos[my_order[i]]; }
    for (uint32_t i : scene_leaf_last) out.tris[i].last = 1;
    out.ref_tris.resize(n);
    out.tri_box.assign((size_t)(n ? n : 1) * 8, 0.f);
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t src = out.figure_order[i];
        out.ref_tris[i] = make(src); out.ref_tris[i].ref_index = i;
        for (int k = 0; k < 3; k++) { out.tri_box[8 * (size_t)i + k] = boxes[src].lo[k]; out.tri_box[8 * (size_t)i + 4 + k] = boxes[src].hi[k]; }
    }
    out.box_c2 = out.box_pad * 0.25f; // 2^-20 x the largest |coordinate|
    out.lights.resize(n_lights);
    for (uint32_t i = 0; i < n_lights; i++) { out.lights[i] = make(out.light_order[i]); out.lights[i].ref_index = i; }
    for (uint32_t i : light_leaf_last) out.lights[i].last = 1;
    // The reference's light tree (constant sort key) is degenerate — thousands of box tests per query — and only the ORDER
    // of its additions matters, which is irrelevant when at most two lights are hit.  A second, properly keyed tree over the
    // same lights serves those queries; three or more hits fall back to the reference-order walk.
    {
        std::vector<uint32_t> forder = out.light_order;
        RefBuilder fast(keys, boxes, forder);
        fast.run(n_lights);
        out.fast_light_bvh_depth = fast.depth;
        std::vector<uint32_t> fast_leaf_last;
        encode_tree(fast.nodes, out.fast_light_nodes, fast_leaf_last, out.box_pad);
        out.fast_lights.resize(n_lights);
        std::vector<uint32_t> light_pos(n, 0);
        for (uint32_t i = 0; i < n_lights; i++) light_pos[out.light_order[i]] = i;
        for (uint32_t i = 0; i < n_lights; i++) { out.fast_lights[i] = make(forder[i]); out.fast_lights[i].ref_index = light_pos[forder[i]]; }
        build_light_sep(light_builder.nodes, n_lights, out.light_sep, out.light_sep_levels);
        for (const RefNode &rn : light_builder.nodes) { out.light_ref.push_back(rn.left); out.light_ref.push_back(rn.right); out.light_ref.push_back(rn.first); out.light_ref.push_back(rn.last); }
        if (out.light_ref.empty()) out.light_ref.assign(4, 0);
        for (uint32_t i : fast_leaf_last) out.fast_lights[i].last = 1;
    }
    out.materials.resize(d.n_materials);
    for (uint32_t i = 0; i < d.n_materials; i++) {
        const rt_material &m = d.materials[i];
        GpuMaterial6 &g = out.materials[i];
        for (int k = 0; k < 3; k++) { g.color[k] = m.base_color[k]; g.emission[k] = m.emission[k]; }
        g.ior = m.ior; g.kind = m.kind;
    }
}

// ---- hw5 --------------------------------------------------------------------------------------------------------
namespace {
struct Q4 { V3 v; float w; };
inline V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 mulk(float k, V3 p) { return {k * p.x, k * p.y, k * p.z}; }
inline float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Q4 qmul(Q4 a, Q4 b) { // quaternion.h:36-38
    return {add(add(mulk(a.w, b.v), mulk(b.w, a.v)), crossr(a.v, b.v)), a.w * b.w - dot3(a.v, b.v)};
}
inline V3 qtransform(Q4 q, V3 p) { // quaternion.h:44-46
    Q4 c{mulk((float)(-1.), q.v), q.w};
    return qmul(qmul(q, Q4{p, 0.f}), c).v;
}
// AABB::AABB(const Figure&), hw5/src/primitives.cpp:171-201
Box3 figure_box5(const rt_primitive &f) {
    V3 mn, mx;
    if (f.type == RT_PRIM_BOX || f.type == RT_PRIM_ELLIPSOID) {
        mn = mulk((float)(-1.), V3{f.data[0], f.data[1], f.data[2]});
        mx = V3{f.data[0], f.data[1], f.data[2]};
    } else {
        mn = {smin(f.data3[0], smin(f.data[0], f.data2[0])), smin(f.data3[1], smin(f.data[1], f.data2[1])), smin(f.data3[2], smin(f.data[2], f.data2[2]))};
        mx = {smax(f.data3[0], smax(f.data[0], f.data2[0])), smax(f.data3[1], smax(f.data[1], f.data2[1])), smax(f.data3[2], smax(f.data[2], f.data2[2]))};
    }
    Q4 q{V3{f.rotation[0], f.rotation[1], f.rotation[2]}, f.rotation[3]};
    Q4 r{mulk((float)(-1.), q.v), q.w}; // rotation.conjugate()
    V3 c0 = qtransform(r, mn);
    Box3 b;
    b.lo[0] = b.hi[0] = c0.x; b.lo[1] = b.hi[1] = c0.y; b.lo[2] = b.hi[2] = c0.z;
    const V3 corners[7] = {{mn.x, mn.y, mx.z}, {mn.x, mx.y, mn.z}, {mn.x, mx.y, mx.z}, {mx.x, mn.y, mn.z}, {mx.x, mn.y, mx.z}, {mx.x, mx.y, mn.z}, {mx.x, mx.y, mx.z}};
    for (const V3 &c : corners) {
        V3 p = qtransform(r, c);
        b.hi[0] = smax(b.hi[0], p.x); b.hi[1] = smax(b.hi[1], p.y); b.hi[2] = smax(b.hi[2], p.z);
        b.lo[0] = smin(b.lo[0], p.x); b.lo[1] = smin(b.lo[1], p.y); b.lo[2] = smin(b.lo[2], p.z);
    }
    for (int k = 0; k < 3; k++) { b.lo[k] = b.lo[k] + f.position[k]; b.hi[k] = b.hi[k] + f.position[k]; }
    return b;
}
} // namespace

void prepare_scene_hw5(const rt_scene_desc &d, PreparedScene5 &out) {
    const uint32_t n = d.n_primitives;
    if (n >= 0x7FFFFFFFu) throw std::runtime_error("too many primitives");
    std::vector<float> keys[3];
    std::vector<Box3> boxes(n);
    for (int k = 0; k < 3; k++) keys[k].resize(n);
    for (uint32_t i = 0; i < n; i++) {
        const rt_primitive &f = d.primitives[i];
        if (f.type < RT_PRIM_ELLIPSOID || f.type > RT_PRIM_TRIANGLE) throw std::runtime_error("bad primitive type");
        for (int k = 0; k < 3; k++) keys[k][i] = f.position[k];   // bvh.h:61-63 sorts on Figure::position
        if (f.type != RT_PRIM_PLANE) boxes[i] = figure_box5(f);
        else memset(&boxes[i], 0, sizeof(Box3));
    }
    // Scene::initBVH, hw5/src/scene.cpp:18-23: planes to the back, BVH (which reorders) over the front
    std::vector<uint32_t> &order = out.figure_order;
    order.resize(n);
    for (uint32_t i = 0; i < n; i++) order[i] = i;
    out.n_nonplanes = (uint32_t)(std::partition(order.begin(), order.end(), [&](uint32_t i) { return d.primitives[i].type != RT_PRIM_PLANE; }) - order.begin());
    RefBuilder scene_builder(keys, boxes, order);
    scene_builder.run(out.n_nonplanes);
    out.bvh_depth = scene_builder.depth;
    // FiguresMix::FiguresMix on a copy of the reordered list, hw5/src/include/distributions.h:180-198
    std::vector<uint32_t> lorder = order;
    uint32_t n_lights = (uint32_t)(std::partition(lorder.begin(), lorder.end(), [&](uint32_t i) {
        const rt_primitive &f = d.primitives[i];
        if (f.emission[0] == 0 && f.emission[1] == 0 && f.emission[2] == 0) return false;
        return f.type == RT_PRIM_BOX || f.type == RT_PRIM_ELLIPSOID || f.type == RT_PRIM_TRIANGLE;
    }) - lorder.begin());
    RefBuilder light_builder(keys, boxes, lorder);
    light_builder.run(n_lights);
    out.light_bvh_depth = light_builder.depth;
    out.light_order.assign(lorder.begin(), lorder.begin() + n_lights);
    std::vector<uint32_t> scene_leaf_last, light_leaf_last;
    encode_tree(scene_builder.nodes, out.nodes, scene_leaf_last);
    encode_tree(light_builder.nodes, out.light_nodes, light_leaf_last);
    encode_ref_tree(scene_builder.nodes, out.ref_nodes);
    encode_ref_tree(light_builder.nodes, out.ref_light_nodes);
    auto make = [&](uint32_t src) {
        const rt_primitive &f = d.primitives[src];
        GpuFig5 g;
        memset(&g, 0, sizeof g);
        for (int k = 0; k < 3; k++) {
            g.data[k] = f.data[k]; g.data2[k] = f.data2[k]; g.data3[k] = f.data3[k]; g.position[k] = f.position[k];
            g.color[k] = f.color[k]; g.emission[k] = f.emission[k];
        }
        for (int k = 0; k < 4; k++) g.rotation[k] = f.rotation[k];
        g.type = f.type; g.kind = f.kind; g.ior = f.ior;
        return g;
    };
    out.figs.resize(n);
    for (uint32_t i = 0; i < n; i++) out.figs[i] = make(order[i]);
    for (uint32_t i : scene_leaf_last) out.figs[i].last = 1;
    out.lights.resize(n_lights);
    for (uint32_t i = 0; i < n_lights; i++) out.lights[i] = make(lorder[i]);
    for (uint32_t i : light_leaf_last) out.lights[i].last = 1;
}

} // namespace rtamd
