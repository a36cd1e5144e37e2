// Scene-tree construction on the GPU (the reference builds on the host: hw8/src/include/bvh.h:34-109, full-sweep SAH
// over three sorted axes, O(n log^2 n)).  This builder is its own design: top-down binned SAH, one tree level per step, all nodes of
// a level in parallel, nothing physically sorted until the end.
//
//   per level   bin      every primitive of a node that is still open adds its box to 3 x 16 bins of that node (centroid bins along
//                        x, y, z over the node's centroid bounds; float min / max as atomicMax on order-preserving integers)
//               split    one thread per open node sweeps the bins: cost(axis, plane) = A(left) n_left + A(right) n_right; the node
//                        becomes a leaf when no plane beats A(node) n (the reference's criterion, bvh.h:92-95) and it holds at most
//                        8 primitives, or when it reached the depth limit; deep nodes that are still large split for balance
//               number   one workgroup: prefix sums give the children their node numbers and bin slots in the order of their parents
//                        (the numbering, and with it the leaf order and every tie between equal hit distances, is deterministic)
//               assign   every primitive of a split node moves to its child and widens the child's centroid bounds
//   at the end  leaves are laid out in node order (prefix sum), their primitives sorted by load index; inner nodes become two-box
//               GpuNodes (padded like scene_prep.cpp pad_box) numbered level by level, i.e. the top of the tree is contiguous.
//
// The tree depth is bounded (BvbView::max_depth), so the LDS stack columns of the persistent kernels always fit.  The traversal result does
// not depend on the tree (closest hit with the tie rule on the reference's figure index where the caller keeps one), which is what makes
// a tree of our own legitimate for hw6's replay (rt_kernels_hw6.h) and for hw8 scenes built with RT_BUILD_DEVICE_BVH.
#pragma once
#include "rt_types.h"
#include <hip/hip_runtime.h>

namespace rtamd {
namespace dev {

#define BVB_BINS 16
#define BVB_BIN_WORDS 7                      // count, ~ord(lo.xyz), ord(hi.xyz)
// per open node: 3 x 16 centroid bins, then two pseudo bins whose count words hold how many of the node's primitives have index-hash
// bit 0 / 1 at the node's depth: the split of last resort when no plane separates the centroids (bvb_hash_bit)
#define BVB_HASH_BINS (3 * BVB_BINS * BVB_BIN_WORDS)
#define BVB_NODE_BIN_WORDS (BVB_HASH_BINS + 2 * BVB_BIN_WORDS)
#define BVB_MAX_DEPTH 28                     // upper bound of BvbView::max_depth (leaves at depth <= max_depth: a traversal stack of that many entries is enough)
#define BVB_MAX_LEAF 8                       // above this a node is split even when the SAH says "leaf"
#define BVB_NONE 0xFFFFFFFFu
#define BVB_THREADS 1024
#define BVB_LDS_NODES 32                     // levels with at most this many open nodes bin through LDS (all primitives hit a few lines otherwise)

struct BvbNode {                             // 64 B
    float lo[3]; uint32_t count;
    float hi[3]; uint32_t left;              // children left, left + 1; 0 = leaf (or still open)
    uint32_t clo[3]; uint32_t info;          // centroid bounds as atomicMax keys (clo of ~ord); info = depth | axis << 8 | plane << 12
    uint32_t chi[3]; uint32_t rank;          // open node: its bin slot; afterwards: leaf = first slot in the leaf order, inner = GpuNode index
};

struct BvbState {
    uint32_t n_nodes, n_open, n_open_next, max_depth, n_inner, n_leaf_slots;
    uint32_t level_base;                     // first node created by the level's number step (its children are level_base .. n_nodes - 1)
};

struct BvbDecision {                         // 64 B, per open node of the level
    float llo[3]; uint32_t split;            // 1 = split at (axis, plane)
    float lhi[3]; uint32_t n_left;
    float rlo[3]; uint32_t axis_plane;
    float rhi[3]; uint32_t n_right;
};

struct BvbView {
    const float *boxes;                      // 8 floats per primitive: lo.xyz, -, hi.xyz, -
    uint32_t n;
    uint32_t *prim_node;
    BvbNode *nodes;
    uint32_t *bins;
    BvbDecision *dec;
    uint32_t *open_cur, *open_next;
    BvbState *st;
    uint32_t *order;                         // leaf slot -> primitive
    GpuNode *out_nodes;
    float abs_pad;                           // absolute part of the padding of the emitted boxes (scene_prep.cpp pad_box)
    uint32_t max_depth;                      // depth limit of this build (<= BVB_MAX_DEPTH): the stack columns of the kernel that will walk the tree
};

__device__ __forceinline__ uint32_t bvb_ord(float f) { uint32_t b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__device__ __forceinline__ float bvb_unord(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); }
__device__ __forceinline__ float bvb_half_area(const float *lo, const float *hi) {
    const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
    return ex * ey + ey * ez + ez * ex;
}
// Bin of a primitive along one axis of its node: twice the centroid (lo + hi) against the node's doubled centroid bounds.
__device__ __forceinline__ int bvb_bin(float c2, float clo2, float chi2) {
    const float e = chi2 - clo2;
    if (!(e > 0.f)) return 0;
    const int b = (int)((c2 - clo2) * ((float)BVB_BINS / e));
    return b < 0 ? 0 : (b > BVB_BINS - 1 ? BVB_BINS - 1 : b);
}

// When all centroids of a node coincide (identical or concentric primitives) no bin plane separates them; a node of 1e5 such primitives
// would become one leaf that a single thread sorts (O(n^2)) and every ray scans.  Then the node is halved by a hash bit of the primitive
// index instead: the children share the node's box, the subtree is balanced, leaves stay small.
__device__ __forceinline__ uint32_t bvb_hash_bit(uint32_t p, uint32_t depth) {
    uint32_t h = p * 0x9E3779B1u; h ^= h >> 15; h *= 0x85EBCA6Bu; h ^= h >> 13;
    return (h >> (depth & 31u)) & 1u;
}

__global__ void bvb_init_kernel(BvbView B) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t key[12]; // root bounds: primitive bounds (kept in the centroid-key fields of node 1, not a node yet) and centroid bounds
    for (int k = 0; k < 12; k++) key[k] = 0;
    if (p < B.n) {
        B.prim_node[p] = 0;
        const float *b = B.boxes + 8 * (size_t)p;
        for (int k = 0; k < 3; k++) {
            key[k] = ~bvb_ord(b[k]); key[3 + k] = bvb_ord(b[4 + k]);
            key[6 + k] = ~bvb_ord(b[k] + b[4 + k]); key[9 + k] = bvb_ord(b[k] + b[4 + k]);
        }
    }
    for (int k = 0; k < 12; k++)
        for (int d = 32; d >= 1; d >>= 1) key[k] = max(key[k], (uint32_t)__shfl_xor((int)key[k], d));
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 3; k++) {
            atomicMax(&B.nodes[1].clo[k], key[k]); atomicMax(&B.nodes[1].chi[k], key[3 + k]);
            atomicMax(&B.nodes[0].clo[k], key[6 + k]); atomicMax(&B.nodes[0].chi[k], key[9 + k]);
        }
}

__global__ void bvb_root_kernel(BvbView B) {
    if (threadIdx.x || blockIdx.x) return;
    BvbNode *root = B.nodes;
    for (int k = 0; k < 3; k++) {
        root->lo[k] = bvb_unord(~B.nodes[1].clo[k]); root->hi[k] = bvb_unord(B.nodes[1].chi[k]);
        B.nodes[1].clo[k] = 0; B.nodes[1].chi[k] = 0;
    }
    root->count = B.n; root->left = 0; root->info = 0;
    const bool open = B.n > 1;
    root->rank = open ? 0u : BVB_NONE;
    B.open_cur[0] = 0;
    B.st->n_nodes = 1; B.st->n_open = open ? 1u : 0u; B.st->n_open_next = 0; B.st->max_depth = 0; B.st->n_inner = 0; B.st->n_leaf_slots = 0; B.st->level_base = 1;
}

__global__ __launch_bounds__(BVB_THREADS) void bvb_bin_kernel(BvbView B) {
    __shared__ uint32_t lbins[BVB_LDS_NODES * BVB_NODE_BIN_WORDS];
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n_open = B.st->n_open;
    if (n_open == 0) return;
    const bool via_lds = n_open <= BVB_LDS_NODES; // uniform: near the root every primitive would hit the same few cache lines
    if (via_lds) {
        for (uint32_t i = threadIdx.x; i < n_open * BVB_NODE_BIN_WORDS; i += blockDim.x) lbins[i] = 0;
        __syncthreads();
    }
    if (p < B.n) {
        const BvbNode *N = B.nodes + B.prim_node[p];
        const uint32_t rank = N->rank;
        if (N->left == 0 && rank != BVB_NONE) {
            const float *b = B.boxes + 8 * (size_t)p;
            const float lo[3] = {b[0], b[1], b[2]}, hi[3] = {b[4], b[5], b[6]};
            uint32_t *bins = (via_lds ? lbins : B.bins) + (size_t)rank * BVB_NODE_BIN_WORDS;
            for (int k = 0; k < 3; k++) {
                const int bin = bvb_bin(lo[k] + hi[k], bvb_unord(~N->clo[k]), bvb_unord(N->chi[k]));
                uint32_t *w = bins + (k * BVB_BINS + bin) * BVB_BIN_WORDS;
                atomicAdd(w, 1u);
                for (int j = 0; j < 3; j++) { atomicMax(w + 1 + j, ~bvb_ord(lo[j])); atomicMax(w + 4 + j, bvb_ord(hi[j])); }
            }
            atomicAdd(bins + BVB_HASH_BINS + bvb_hash_bit(p, N->info & 255u) * BVB_BIN_WORDS, 1u);
        }
    }
    if (via_lds) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n_open * BVB_NODE_BIN_WORDS; i += blockDim.x) {
            const uint32_t v = lbins[i];
            if (v) { if (i % BVB_BIN_WORDS == 0) atomicAdd(B.bins + i, v); else atomicMax(B.bins + i, v); }
        }
    }
}

__global__ void bvb_split_kernel(BvbView B) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= B.st->n_open) return;
    const uint32_t node = B.open_cur[r];
    BvbNode *N = B.nodes + node;
    const uint32_t count = N->count, depth = N->info & 255u;
    uint32_t *bins = B.bins + (size_t)r * BVB_NODE_BIN_WORDS;
    const float inf = 3.0e38f;
    float best = inf; int best_axis = -1, best_plane = 0;
    float bal_best = inf; int bal_axis = -1, bal_plane = 0;
    for (int k = 0; k < 3; k++) {
        const uint32_t *w = bins + k * BVB_BINS * BVB_BIN_WORDS;
        float r_area[BVB_BINS]; uint32_t r_cnt[BVB_BINS];  // suffix: bins i .. 15
        float lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
        uint32_t c = 0;
        for (int i = BVB_BINS - 1; i >= 1; i--) {
            const uint32_t ci = w[i * BVB_BIN_WORDS];
            if (ci) for (int j = 0; j < 3; j++) { lo[j] = fminf(lo[j], bvb_unord(~w[i * BVB_BIN_WORDS + 1 + j])); hi[j] = fmaxf(hi[j], bvb_unord(w[i * BVB_BIN_WORDS + 4 + j])); }
            c += ci;
            r_cnt[i] = c; r_area[i] = c ? bvb_half_area(lo, hi) : 0.f;
        }
        for (int j = 0; j < 3; j++) { lo[j] = inf; hi[j] = -inf; }
        c = 0;
        for (int i = 0; i < BVB_BINS - 1; i++) { // plane i: bins 0..i | i+1..15
            const uint32_t ci = w[i * BVB_BIN_WORDS];
            if (ci) for (int j = 0; j < 3; j++) { lo[j] = fminf(lo[j], bvb_unord(~w[i * BVB_BIN_WORDS + 1 + j])); hi[j] = fmaxf(hi[j], bvb_unord(w[i * BVB_BIN_WORDS + 4 + j])); }
            c += ci;
            const uint32_t cr = r_cnt[i + 1];
            if (c == 0 || cr == 0) continue;
            const float cost = bvb_half_area(lo, hi) * (float)c + r_area[i + 1] * (float)cr;
            if (cost < best) { best = cost; best_axis = k; best_plane = i; }
            const float bal = fabsf((float)c - (float)cr);
            if (bal < bal_best) { bal_best = bal; bal_axis = k; bal_plane = i; }
        }
    }
    bool split = best_axis >= 0;
    const uint32_t h0 = bins[BVB_HASH_BINS], h1 = bins[BVB_HASH_BINS + BVB_BIN_WORDS];
    const bool by_hash = !split && count > BVB_MAX_LEAF && depth < B.max_depth && h0 > 0 && h1 > 0; // no plane separates the centroids
    if (split && count <= BVB_MAX_LEAF && !(best < bvb_half_area(N->lo, N->hi) * (float)count)) split = false; // bvh.h:92-95
    if (depth >= B.max_depth) split = false;
    // balance guard: a node too large for the levels that are left below it splits where the counts are most even
    if (split && depth + 2 < B.max_depth && ((unsigned long long)count * 4ull > (1ull << (B.max_depth - depth - 2)) * (unsigned long long)BVB_MAX_LEAF)) { best_axis = bal_axis; best_plane = bal_plane; }
    BvbDecision D;
    D.split = split ? 1u : 0u; D.axis_plane = 0; D.n_left = 0; D.n_right = 0;
    for (int j = 0; j < 3; j++) { D.llo[j] = 0.f; D.lhi[j] = 0.f; D.rlo[j] = 0.f; D.rhi[j] = 0.f; }
    if (split) {
        const uint32_t *w = bins + best_axis * BVB_BINS * BVB_BIN_WORDS;
        float llo[3] = {inf, inf, inf}, lhi[3] = {-inf, -inf, -inf}, rlo[3] = {inf, inf, inf}, rhi[3] = {-inf, -inf, -inf};
        uint32_t nl = 0, nr = 0;
        for (int i = 0; i < BVB_BINS; i++) {
            const uint32_t ci = w[i * BVB_BIN_WORDS];
            if (!ci) continue;
            const bool left = i <= best_plane;
            for (int j = 0; j < 3; j++) {
                const float l = bvb_unord(~w[i * BVB_BIN_WORDS + 1 + j]), h = bvb_unord(w[i * BVB_BIN_WORDS + 4 + j]);
                if (left) { llo[j] = fminf(llo[j], l); lhi[j] = fmaxf(lhi[j], h); } else { rlo[j] = fminf(rlo[j], l); rhi[j] = fmaxf(rhi[j], h); }
            }
            if (left) nl += ci; else nr += ci;
        }
        for (int j = 0; j < 3; j++) { D.llo[j] = llo[j]; D.lhi[j] = lhi[j]; D.rlo[j] = rlo[j]; D.rhi[j] = rhi[j]; }
        D.n_left = nl; D.n_right = nr; D.axis_plane = (uint32_t)best_axis | ((uint32_t)best_plane << 4);
    }
    if (by_hash) { // both halves keep the node's box
        D.split = 1u; D.axis_plane = 3u; D.n_left = h0; D.n_right = h1;
        for (int j = 0; j < 3; j++) { D.llo[j] = N->lo[j]; D.lhi[j] = N->hi[j]; D.rlo[j] = N->lo[j]; D.rhi[j] = N->hi[j]; }
    }
    B.dec[r] = D;
    for (int i = 0; i < BVB_NODE_BIN_WORDS; i++) bins[i] = 0; // the slot is clean for the next level
}

// Exclusive prefix sums of two values over the workgroup (1024 threads = 16 waves); `carry` continues a running total.
__device__ __forceinline__ void bvb_scan2(uint32_t a, uint32_t b, uint32_t &ea, uint32_t &eb, uint32_t &ta, uint32_t &tb, uint32_t (*lds)[2]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t ia = a, ib = b;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t ua = __shfl_up(ia, d), ub = __shfl_up(ib, d);
        if (lane >= d) { ia += ua; ib += ub; }
    }
    __syncthreads();
    if (lane == 63) { lds[wave][0] = ia; lds[wave][1] = ib; }
    __syncthreads();
    uint32_t wa = 0, wb = 0; ta = 0; tb = 0;
    for (int w = 0; w < BVB_THREADS / 64; w++) { if (w < wave) { wa += lds[w][0]; wb += lds[w][1]; } ta += lds[w][0]; tb += lds[w][1]; }
    ea = wa + ia - a; eb = wb + ib - b;
}

__global__ __launch_bounds__(BVB_THREADS) void bvb_number_kernel(BvbView B) {
    __shared__ uint32_t lds[BVB_THREADS / 64][2];
    const uint32_t n_open = B.st->n_open, base_node = B.st->n_nodes;
    uint32_t run_split = 0, run_open = 0;
    for (uint32_t r0 = 0; r0 < n_open; r0 += BVB_THREADS) {
        const uint32_t r = r0 + threadIdx.x;
        uint32_t s = 0, o = 0, node = 0, depth = 0;
        BvbDecision D;
        bool ol = false, orr = false;
        if (r < n_open) {
            D = B.dec[r];
            node = B.open_cur[r];
            depth = B.nodes[node].info & 255u;
            s = D.split;
            if (s) { ol = D.n_left > 1; orr = D.n_right > 1; o = (ol ? 1u : 0u) + (orr ? 1u : 0u); }
        }
        uint32_t es, eo, ts, to;
        bvb_scan2(s, o, es, eo, ts, to, lds);
        if (r < n_open) {
            BvbNode *N = B.nodes + node;
            N->rank = BVB_NONE;
            if (s) {
                const uint32_t left = base_node + 2u * (run_split + es);
                N->left = left;
                N->info = depth | ((D.axis_plane & 15u) << 8) | ((D.axis_plane >> 4) << 12);
                uint32_t slot = run_open + eo;
                BvbNode L, R;
                for (int k = 0; k < 3; k++) { L.lo[k] = D.llo[k]; L.hi[k] = D.lhi[k]; R.lo[k] = D.rlo[k]; R.hi[k] = D.rhi[k]; L.clo[k] = 0; L.chi[k] = 0; R.clo[k] = 0; R.chi[k] = 0; }
                L.count = D.n_left; R.count = D.n_right; L.left = 0; R.left = 0; L.info = depth + 1; R.info = depth + 1;
                L.rank = BVB_NONE; R.rank = BVB_NONE;
                if (ol) { L.rank = slot; B.open_next[slot] = left; slot++; }
                if (orr) { R.rank = slot; B.open_next[slot] = left + 1u; }
                B.nodes[left] = L; B.nodes[left + 1u] = R;
            }
        }
        run_split += ts; run_open += to;
    }
    __syncthreads();
    if (threadIdx.x == 0) { B.st->n_nodes = base_node + 2u * run_split; B.st->n_open_next = run_open; B.st->level_base = base_node; }
}

__global__ __launch_bounds__(BVB_THREADS) void bvb_assign_kernel(BvbView B) {
    __shared__ uint32_t lkeys[2 * BVB_LDS_NODES * 6];
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n_open = B.st->n_open, base = B.st->level_base, n_children = B.st->n_nodes - base;
    if (n_open == 0) return;
    const bool via_lds = n_open <= BVB_LDS_NODES;
    if (via_lds) {
        for (uint32_t i = threadIdx.x; i < n_children * 6; i += blockDim.x) lkeys[i] = 0;
        __syncthreads();
    }
    if (p < B.n) {
        const BvbNode *N = B.nodes + B.prim_node[p];
        if (N->left != 0) {
            const int axis = (int)((N->info >> 8) & 15u), plane = (int)(N->info >> 12);
            const float *b = B.boxes + 8 * (size_t)p;
            uint32_t child;
            if (axis == 3) child = N->left + bvb_hash_bit(p, N->info & 255u); // the split of last resort
            else {
                const int bin = bvb_bin(b[axis] + b[4 + axis], bvb_unord(~N->clo[axis]), bvb_unord(N->chi[axis]));
                child = N->left + (bin > plane ? 1u : 0u);
            }
            B.prim_node[p] = child;
            BvbNode *C = B.nodes + child;
            if (C->count > 1) {
                uint32_t *klo = via_lds ? lkeys + (child - base) * 6 : C->clo, *khi = via_lds ? klo + 3 : C->chi;
                for (int k = 0; k < 3; k++) { atomicMax(klo + k, ~bvb_ord(b[k] + b[4 + k])); atomicMax(khi + k, bvb_ord(b[k] + b[4 + k])); }
            }
        }
    }
    if (via_lds) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n_children * 6; i += blockDim.x) {
            const uint32_t v = lkeys[i];
            if (v) { BvbNode *C = B.nodes + base + i / 6; atomicMax((i % 6) < 3 ? &C->clo[i % 6] : &C->chi[i % 6 - 3], v); }
        }
    }
}

__global__ void bvb_next_level_kernel(BvbView B) { // the lists were swapped by the host
    if (threadIdx.x || blockIdx.x) return;
    B.st->n_open = B.st->n_open_next; B.st->n_open_next = 0;
}

// ---- layout -----------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BVB_THREADS) void bvb_layout_kernel(BvbView B) { // leaf slots and GpuNode indices in node order
    __shared__ uint32_t lds[BVB_THREADS / 64][2];
    const uint32_t n_nodes = B.st->n_nodes;
    uint32_t run_leaf = 0, run_inner = 0, max_depth = 0;
    for (uint32_t i0 = 0; i0 < n_nodes; i0 += 4 * BVB_THREADS) { // four consecutive nodes per thread and step
        const uint32_t first = i0 + 4u * threadIdx.x;
        uint32_t a = 0, b = 0, cnt[4], leaf = 0;
        for (uint32_t j = 0; j < 4; j++) {
            cnt[j] = 0;
            if (first + j < n_nodes) {
                const BvbNode *N = B.nodes + first + j;
                if (N->left == 0) { cnt[j] = N->count; a += cnt[j]; leaf |= 1u << j; max_depth = max(max_depth, N->info & 255u); } else b++;
            }
        }
        uint32_t ea, eb, ta, tb;
        bvb_scan2(a, b, ea, eb, ta, tb, lds);
        for (uint32_t j = 0; j < 4; j++)
            if (first + j < n_nodes) {
                BvbNode *N = B.nodes + first + j;
                if ((leaf >> j) & 1u) { N->rank = run_leaf + ea; ea += cnt[j]; } else { N->rank = run_inner + eb; eb++; }
                N->clo[0] = 0;
            }
        run_leaf += ta; run_inner += tb;
    }
    atomicMax(&B.st->max_depth, max_depth);
    if (threadIdx.x == 0) { B.st->n_inner = run_inner; B.st->n_leaf_slots = run_leaf; }
}

__global__ void bvb_place_kernel(BvbView B) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= B.n) return;
    BvbNode *N = B.nodes + B.prim_node[p];
    B.order[N->rank + atomicAdd(&N->clo[0], 1u)] = p;
}

__device__ __forceinline__ void bvb_pad_box(const float *lo, const float *hi, float *olo, float *ohi, float abs_pad) { // scene_prep.cpp pad_box
    for (int k = 0; k < 3; k++) {
        const float mag = fmaxf(fabsf(lo[k]), fabsf(hi[k]));
        const float pad = mag * 7.62939453125e-06f + abs_pad + 1e-30f;
        olo[k] = lo[k] - pad; ohi[k] = hi[k] + pad;
    }
}

// One thread per node: a leaf sorts its slots by load index; an inner node writes its GpuNode.
__global__ void bvb_emit_kernel(BvbView B) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n_nodes = B.st->n_nodes;
    if (i >= n_nodes) return;
    const BvbNode *N = B.nodes + i;
    if (N->left == 0) {
        uint32_t *o = B.order + N->rank;
        for (uint32_t a = 1; a < N->count; a++) {
            const uint32_t v = o[a];
            uint32_t j = a;
            while (j > 0 && o[j - 1] > v) { o[j] = o[j - 1]; j--; }
            o[j] = v;
        }
        if (n_nodes == 1) { // the root is a leaf: wrap it (scene_prep.cpp encode_tree)
            GpuNode g;
            bvb_pad_box(N->lo, N->hi, g.lo0, g.hi0, B.abs_pad);
            g.child0 = (int32_t)(0x80000000u | 0u); g.cnt0 = (int32_t)N->count;
            for (int k = 0; k < 3; k++) { g.lo1[k] = 3.0e38f; g.hi1[k] = 3.0e38f; }
            g.child1 = (int32_t)0xFFFFFFFFu; g.cnt1 = 0;
            if (N->count == 0) { for (int k = 0; k < 3; k++) { g.lo0[k] = 3.0e38f; g.hi0[k] = 3.0e38f; } g.child0 = (int32_t)0xFFFFFFFFu; g.cnt0 = 0; }
            B.out_nodes[0] = g;
        }
        return;
    }
    const BvbNode *L = B.nodes + N->left, *R = L + 1;
    GpuNode g;
    bvb_pad_box(L->lo, L->hi, g.lo0, g.hi0, B.abs_pad);
    bvb_pad_box(R->lo, R->hi, g.lo1, g.hi1, B.abs_pad);
    if (L->left == 0) { g.child0 = (int32_t)(0x80000000u | L->rank); g.cnt0 = (int32_t)L->count; } else { g.child0 = (int32_t)L->rank; g.cnt0 = 0; }
    if (R->left == 0) { g.child1 = (int32_t)(0x80000000u | R->rank); g.cnt1 = (int32_t)R->count; } else { g.child1 = (int32_t)R->rank; g.cnt1 = 0; }
    B.out_nodes[N->rank] = g;
}

// last[slot] = 1 for the last slot of every leaf (the kernels walk a leaf until they meet that mark)
__global__ void bvb_marks_kernel(BvbView B, uint8_t *last) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B.st->n_nodes) return;
    const BvbNode *N = B.nodes + i;
    if (N->left == 0 && N->count > 0) last[N->rank + N->count - 1u] = 1;
}

// ---- gathers into the leaf order -------------------------------------------------------------------------------------------------------
template <class T> __global__ void bvb_gather_kernel(const T *in, T *out, const uint32_t *order, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[order[i]];
}

} // namespace dev
} // namespace rtamd
