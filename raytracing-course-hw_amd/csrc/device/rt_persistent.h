// Persistent dataflow form of the hw8 replay path tracer: ONE launch renders the frame, no rounds and no global barrier.
//
// The reference's per-pixel loop (hw8/src/scene.cpp:84-177) is a chain of dependent stages per pixel — closest hit
// (bvh.h:111-142) -> shade / sample (scene.cpp:99-156) -> light-pdf sum (distributions.h:148-165) -> throughput update
// (scene.cpp:158-164) -> next bounce or next sample — and the chain of one pixel never depends on another pixel's.  The
// round-based pipeline of rt_wavefront.h runs each stage as a kernel over all pixels and pays a machine-wide drain at every
// kernel boundary (the launch ends with its longest walk).  Here the dependency is kept PER PATH:
//
//   * five 4-wave workgroups per CU (one wave per SIMD each: five waves per SIMD at <= 96 VGPRs) each own a fixed, interleaved
//     share of the pixel slots (8x8 sub-tiles dealt round-robin to the workgroups); the path records stay in HBM in the layout
//     of rt_wavefront.h, but only this workgroup touches them, so hand-offs between its waves need workgroup-scope ordering
//     only (one CU, one L1: `s_waitcnt` + LDS);
//   * the stage a path waits for is one bit per path in an LDS bitmap (need_trace / need_light / need_shade / ...); a wave
//     that wants work claims set bits with LDS atomics (one word per lane, rotating cursor: round-robin service, no
//     capacity limit, no ring to overflow);
//   * every wave picks a role when it is idle: closest-hit walker, light-sum walker or shader, by the populations of the
//     bitmaps; walkers keep their lanes full by refilling idle lanes from the bitmap; a shader takes up to 64 paths,
//     finishes their pending bounce, shades the new hit and sets the bits of what each path needs next;
//   * the walkers read four-wide nodes on a 16-bit grid (rt_types.h GpuNode4Q: two tree levels per 64-byte fetch, the ray in grid
//     coordinates), park the leaves they meet for a common leaf phase, and leave what only a few lanes need (a light hit's pdf
//     term, the record of a walk that has ended) for once per pass;
//   * the speculative pairing of rt_wavefront.h is kept: a bounce's sampled direction is traced for the next hit at the
//     same time as its light-pdf sum is walked; a 2-bit "pending" field per path joins the two (whoever finishes last sets
//     need_shade).
//
// What this removes: 3 x spp x depth kernel launches, the global queue atomics, the per-round drain (every launch of the round
// pipeline waited for its longest walk), and the idle memory system during traversal / idle ALUs during shading (the roles
// overlap on every CU).  A path advances as fast as its own chain allows, which is what lets a small frame (a shard of a
// multi-GPU render) run near the full rate.
//
// Reference-exact box decisions (hw8/src/primitives.cpp:29-53,163-165).  The walkers prune with a cheap conservative test on
// padded boxes, so they find a superset of the triangles the reference's own slab test lets through.  A hit is accepted
// as it stands when the hit point lies robustly inside its triangle's box (pt_box_robust: then every ancestor box passes the
// reference's test whatever the rounding) and no second triangle was hit within a few ulp of it; the rare others are walked
// again by the `exact` role with the reference's arithmetic on the unpadded boxes of the reference tree (ref_closest_hit,
// ref_light_pdf_sum).  Pixels then match the reference's also where a ray grazes a box corner.  Rays that cross a tripwire (rt_exact.h
// pt_tripwire: the leaf box of a triangle whose test accepts points far away from it) never go to the walkers at all.
#pragma once
#include "rt_wavefront.h"

namespace rtamd {
namespace dev {

// hw8 / hw7: workgroups of FOUR waves (one per SIMD), FIVE resident per CU = five waves per SIMD.  That takes <= 96 VGPRs per wave (no
// scratch), 24-entry stack columns (4 x 24 x 256 B = 24 KB per workgroup) and 1.4 bytes of LDS per path for bitmaps and group tables: 31.7 KB per
// workgroup, five of which fill the CU's 160 KB (handed out in 1,280-byte granules: 25 granules each).
#define P8_WAVES 4
#define P8_THREADS (64 * P8_WAVES)
#ifndef P8_PER_CU
#define P8_PER_CU 5
#endif
#define P8_STACK 24                   // LDS traversal stack entries per lane; the walkers' tree is built at most this deep (rt_bvh_build.h)
#ifndef PT_MAX_PATHS
#define PT_MAX_PATHS 5120
#endif
#define PT_MAX_PATHS_NOTE             // paths per workgroup (bitmap capacity in LDS): x 1,280 workgroups = 6.5 M (a 3840x2160 frame on one GPU: two passes)
#define PT_MIN_GROUP 16               // smallest group of the deal (group_shift 4): the LDS tables are sized for it
#define PT_NW (PT_MAX_PATHS / 32)
#define PT_BIT_T 1u                   // pending: closest-hit walk outstanding
#define PT_BIT_L 2u                   // pending: light-pdf sum outstanding
#define PT_NONE 0xFFFFFFFFu
// indices into PtShared::need / PtShared::cnt
#define PT_Q_TRACE 0
#define PT_Q_LIGHT 1
#define PT_Q_SHADE 2
#define PT_Q_XLIGHT 3                 // light sums for the exact walk (more than WF_MAX_LIGHT_HITS hits, or a hit at a box boundary)
#define PT_Q_XTRACE 4                 // closest hits for the exact walk
#define PT_N_LIVE 5                   // cnt only: pixels of this workgroup not finished yet
#define PT_W_TRACE 6                  // cnt only: waves currently walking closest hits / light sums
#define PT_W_LIGHT 7
#define PT_GSHIFT 15                  // cnt only: PtParams::group_shift (constant during the launch)
// What a sub-tile costs its workgroup, in units of one closest-hit node step (wave time by role over steps by role on the benchmark
// scene): the measure the frame is re-dealt by after its first phase.  Counting shaded hits alone misses the rays that hit nothing.
#ifndef PT_COST_TRACE_NODE
#define PT_COST_TRACE_NODE 1u     // a node visit of a closest-hit walk
#endif
#define PT_COST_TRACE_TRI 1u      // a triangle test of a closest-hit walk
#ifndef PT_COST_LIGHT_NODE
#define PT_COST_LIGHT_NODE 2u     // a node visit of a light-sum walk
#endif
#define PT_COST_LIGHT_TEST 2u     // a light test of a light-sum walk
#define PT_COST_SHADE 14u
#define PT_DEBUG_BLOCKS 2048           // RTAMD_DEBUG_COUNTERS: workgroups whose start / exit times are recorded (>= 256 CUs x 5)
#ifndef PT_QUANT_NODES
#define PT_QUANT_NODES 1       // the walkers read the four-wide grid nodes (rt_types.h GpuNode4Q: two levels per fetch); 0 = the two-box float nodes, for A/B builds
#endif
#ifndef PT_POSTPONE
#define PT_POSTPONE 1          // walkers keep a leaf they meet for the next leaf phase and walk on (0: they wait at it), for A/B builds
#endif
#if !PT_QUANT_NODES
#undef PT_POSTPONE
#define PT_POSTPONE 0          // the two-box float path keeps the plain loop
#endif
#ifndef PT_PEND_SLOTS
#define PT_PEND_SLOTS 3        // leaves a lane may hold for the next leaf phase (2 or 3)
#endif
#if PT_PEND_SLOTS == 3
#define PT_LAST_SLOT pend3
#else
#define PT_LAST_SLOT pend2
#endif
#define PT_DRAINED 0xFFFFFFFEu        // `cur` of a lane whose stack is empty and whose last leaf is still to be tested (reads as a leaf: the lane waits)
#define PT_T_OVERFLOW (-1.f)          // t of a closest-hit record whose walk ran out of stack: the exact role redoes the query
// ---- one step of a walk over the four-wide grid nodes (rt_types.h GpuNode4Q), shared with rt_persistent_hw6.h ------------------------
#define PT_WIDE_NONE 0                // no child entered: the caller pops its stack
#define PT_WIDE_WENT 1                // `cur` is the child to visit next, the others wait in the lane's stack column
#define PT_WIDE_FULL 2                // the column cannot take the children that would wait (nothing was pushed, `cur` unchanged)
// sort key of a child's entry distance: the bits of max(t, 0) (a negative float, -0 included, is a negative integer)
RT_DEV uint32_t pt_near_key(float t) { const int b = (int)__float_as_uint(t); return (uint32_t)(b > 0 ? b : 0); }
// compare-exchange of (key, child word) pairs: the smaller key first
RT_DEV void pt_order(uint32_t &ka, uint32_t &ca, uint32_t &kb, uint32_t &cb) {
    const bool s = kb < ka;
    const uint32_t k = s ? kb : ka, c = s ? cb : ca;
    kb = s ? ka : kb; cb = s ? ca : cb;
    ka = k; ca = c;
}
// Closest-hit walks: the entered children nearest first — the nearest becomes `cur`, the others are pushed farthest first.
// `cap`: entries the column may hold.
RT_DEV int pt_wide_step_nearest(const GpuNode4Q *nodes, const RayGrid &ray, float cull_t, uint32_t (*stack)[64], int lane, int &sp, int cap, uint32_t &cur) {
    const uint4 *q = reinterpret_cast<const uint4 *>(nodes + cur);
    const uint4 b0 = q[0], b1 = q[1], b2 = q[2], b3 = q[3];
    float n0, n1, n2, n3;
    const bool h0 = slab_test_q(b0, ray, cull_t, n0), h1 = slab_test_q(b1, ray, cull_t, n1);
    const bool h2 = slab_test_q(b2, ray, cull_t, n2), h3 = slab_test_q(b3, ray, cull_t, n3);
    uint32_t k0 = h0 ? pt_near_key(n0) : 0xFFFFFFFFu, k1 = h1 ? pt_near_key(n1) : 0xFFFFFFFFu;
    uint32_t k2 = h2 ? pt_near_key(n2) : 0xFFFFFFFFu, k3 = h3 ? pt_near_key(n3) : 0xFFFFFFFFu;
    uint32_t c0 = b0.w, c1 = b1.w, c2 = b2.w, c3 = b3.w;
#ifndef PT_WIDE_SORT
#define PT_WIDE_SORT 1
#endif
#if PT_WIDE_SORT
    pt_order(k0, c0, k1, c1); pt_order(k2, c2, k3, c3); pt_order(k0, c0, k2, c2); pt_order(k1, c1, k3, c3); pt_order(k1, c1, k2, c2);
#else
    pt_order(k0, c0, k1, c1); pt_order(k2, c2, k3, c3); pt_order(k0, c0, k2, c2); // only the nearest is singled out; the others wait in any order
#endif
    const int nh = (int)h0 + (int)h1 + (int)h2 + (int)h3;
    if (nh == 0) return PT_WIDE_NONE;
    if (sp + nh - 1 > cap) return PT_WIDE_FULL;
#if PT_WIDE_SORT
    if (nh > 3) stack[sp++][lane] = c3;
    if (nh > 2) stack[sp++][lane] = c2;
    if (nh > 1) stack[sp++][lane] = c1;
#else
    if (k3 != 0xFFFFFFFFu) stack[sp++][lane] = c3;
    if (k2 != 0xFFFFFFFFu) stack[sp++][lane] = c2;
    if (k1 != 0xFFFFFFFFu) stack[sp++][lane] = c1;
#endif
    cur = c0;
    return PT_WIDE_WENT;
}
// Light sums: every entered child is walked, in any order (the callers keep their hits sorted): the first one now, the others wait.
RT_DEV int pt_wide_step_all(const GpuNode4Q *nodes, const RayGrid &ray, uint32_t (*stack)[64], int lane, int &sp, int cap, uint32_t &cur) {
    const uint4 *q = reinterpret_cast<const uint4 *>(nodes + cur);
    const uint4 b0 = q[0], b1 = q[1], b2 = q[2], b3 = q[3];
    float n;
    const bool h0 = slab_test_q(b0, ray, RT_T_MAX, n), h1 = slab_test_q(b1, ray, RT_T_MAX, n);
    const bool h2 = slab_test_q(b2, ray, RT_T_MAX, n), h3 = slab_test_q(b3, ray, RT_T_MAX, n);
    const int nh = (int)h0 + (int)h1 + (int)h2 + (int)h3;
    if (nh == 0) return PT_WIDE_NONE;
    if (sp + nh - 1 > cap) return PT_WIDE_FULL;
    const int first = h0 ? 0 : h1 ? 1 : h2 ? 2 : 3;
    if (h3 && first != 3) stack[sp++][lane] = b3.w;
    if (h2 && first < 2) stack[sp++][lane] = b2.w;
    if (h1 && first < 1) stack[sp++][lane] = b1.w;
    cur = first == 0 ? b0.w : first == 1 ? b1.w : first == 2 ? b2.w : b3.w;
    return PT_WIDE_WENT;
}
#if PT_QUANT_NODES
typedef RayGrid PtRay;
#define PT_RAY_IDLE RT_GRID_RAY_IDLE
RT_DEV PtRay pt_make_ray(const SceneView &S, F3 o, F3 d) { return make_ray_grid(S.grid, o, d); }
#else
typedef RayInv PtRay;
#define PT_RAY_IDLE {{0.f, 0.f, 0.f}, {1.f, 1.f, 1.f}}
RT_DEV PtRay pt_make_ray(const SceneView &, F3 o, F3 d) { return make_ray_inv(o, d); }
#endif
#define PT_EXACT_BATCH 16             // the exact role walks at most this many queries at once: their stacks (RT_STACK_SIZE entries each) share the wave's LDS stack area

struct PtShared {
    uint32_t stack[P8_WAVES][P8_STACK][64];   // per-lane traversal stack columns, one area per wave
    uint32_t need[5][PT_NW];
    uint32_t pending[PT_NW * 2];              // 2 bits per path
    uint32_t groups[PT_MAX_PATHS / PT_MIN_GROUP]; // local group -> group of the pass
    uint32_t cost[PT_MAX_PATHS / PT_MIN_GROUP];       // work done for each local group in this launch (PT_COST_*): the load measure the frame is re-dealt by
    int cnt[16];
};
static_assert(PT_EXACT_BATCH * RT_STACK_SIZE <= P8_STACK * 64, "the exact role's stacks must fit the wave's LDS stack area");
static_assert(sizeof(PtShared) <= (128 / P8_PER_CU) * 1280, "P8_PER_CU workgroups per CU: the CU's 128 LDS granules of 1,280 bytes shared evenly");

struct PtParams {
    uint32_t n_groups;                // groups of this pass: 2^group_shift consecutive path slots each
    uint32_t group_shift;             // 6: a group is an 8x8 sub-tile; 4: two rows of one (small frames: more, smaller units for the deal — a
                                      // workgroup should hold well over a dozen, and a heavy 8x8 sub-tile alone can outweigh a workgroup's fair share)
    // Which groups a workgroup owns: group_ids[group_ofs[b] .. group_ofs[b + 1]) when group_ofs is set (the host's re-deal after
    // the first phase of a frame), else b, b + n_blocks, b + 2 n_blocks, ...; `resume` = the paths carry on from their records
    // (a later phase) instead of being seeded; group_cost[g] receives the number of hits shaded for group g in this launch.
    const uint32_t *group_ofs, *group_ids;
    uint32_t *group_cost;
    uint32_t resume;
    int refill, leaf_batch;           // as in rt_wavefront.h (refill = closest-hit walker's | light walker's << 16; leaf_batch = batch | share << 16)
    int shade_min;                    // a wave turns shader when this many paths wait for shading (64 = a full wave of them)
    int shade_thr0, shade_thr_step;   // wave w stops refilling its walkers when need_shade holds >= thr0 + w * step paths
    int cost_t, cost_l;               // relative cost of a closest-hit / light query (walker split)
    uint32_t front_first;             // 1: the queues serve the front of the workgroup's group list first (the host sorted it by cost, most expensive first)
    int prio;                         // experiment: 1 = walker stints run at raised wave priority (s_setprio 2), 2 = shader batches do
    unsigned long long deadline_ticks; // 100 MHz ticks a wave may spend in this launch before it gives up (error)
    unsigned long long *counters;     // [0] closest-hit queries, [1] light queries, [2] node visits, [3] triangle tests, [10] discarded speculative hits, [12] exact closest hits, [13] exact light sums, [14] waves that gave up waiting for a lost path (error), [29] waves that ran into the launch deadline (error)
    unsigned long long *debug;        // nullable: per workgroup {start time, exit time of its last wave (100 MHz ticks), paths}
    // COUNT builds, RTAMD_TRACE_PIXEL: every hit record the shader consumes for pixel trace_pixel (= y * width + x) is appended as
    // four float4 (r0..r3 of the path record: ray, hit, packed word); word 0 of trace_buf counts the entries
    float4 *trace_buf; uint32_t trace_cap; int32_t trace_pixel;
};

// COUNT builds only: where a wave's time goes (shader-clock cycles per role) and how full its walker iterations are
struct PtProf {
    unsigned long long t_trace = 0, t_light = 0, t_shade = 0, t_exact = 0, t_idle = 0;
    unsigned long long trace_iters = 0, trace_lane_iters = 0, light_iters = 0, light_lane_iters = 0, stints = 0, shade_batches = 0, shade_items = 0;
    // where a walker's wave time goes (counting build): [0] hand-off and refill, [1] inner nodes, [2] leaves; tests / lane-tests of the leaf loops; light hits
    unsigned long long t_part[2][3] = {{0, 0, 0}, {0, 0, 0}}, leaf_iters[2] = {0, 0}, leaf_lane_iters[2] = {0, 0}, light_hits = 0, light_tests = 0; // the last two per lane
    unsigned long long t_sub[2][3] = {{0, 0, 0}, {0, 0, 0}}, refills[2] = {0, 0}; // of [0]: publish finished walks | take new ones from the bitmap | read their rays
};
template <bool COUNT> struct PtLap { // s_memtime laps of the counting build
    unsigned long long t;
    RT_DEV PtLap() : t(COUNT ? __builtin_amdgcn_s_memtime() : 0ull) {}
    RT_DEV void lap(unsigned long long &acc) { if (COUNT) { const unsigned long long n = __builtin_amdgcn_s_memtime(); acc += n - t; t = n; } }
};

// wave-uniform state
struct PtWave {
    uint32_t nw, n_local, n_blocks, block;
    bool front_first;                 // pt_pop's from_start for the trace / light / shade queues (PtParams::front_first)
    uint32_t cur[5];
};

// A workgroup's paths come in groups of 2^shift consecutive slots (PtParams::group_shift: 6 = an 8x8 sub-tile, 4 = two rows of one): the unit of the deal.
template <class SH> RT_DEV uint32_t pt_gshift(const SH &sh) { return (uint32_t)__builtin_amdgcn_readfirstlane(__hip_atomic_load(&sh.cnt[PT_GSHIFT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)); }
template <class SH> RT_DEV uint32_t pt_slot(const SH &sh, uint32_t l) { const uint32_t g = pt_gshift(sh); return (sh.groups[l >> g] << g) | (l & ((1u << g) - 1u)); }
// A fresh LDS read each time; every lane reads the same word, and readfirstlane makes that explicit: the scheduler's decisions are
// taken on SGPRs (scalar branches, wave-uniform by construction — the code under them uses __ballot / __shfl / lane-0 atomics).
RT_DEV int pt_count(const int *p) { return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)); }

// Lanes that must wait at a leaf before the wave runs its triangle tests: a share (leaf_batch >> 16, in 1/256) of the active lanes, at most
// leaf_batch & 255.  Plain integer arithmetic on purpose: min() of an int and __popcll's result picks the double overload.
RT_DEV int pt_leaf_batch(int leaf_batch, unsigned long long m_active) {
    const int cap = leaf_batch & 255, share = ((int)__popcll(m_active) * (leaf_batch >> 16) + 255) >> 8;
    return share < cap ? share : cap;
}

// The wave's mask of a predicate, straight from the compare (HIP's __ballot takes an int: a select and a second compare per call).
RT_DEV unsigned long long pt_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// Number of set bits of a wave mask below this lane (v_mbcnt: no 64-bit lane mask in registers).
RT_DEV uint32_t pt_rank_below(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// Exclusive prefix sum over the wave of a small count per lane (0..63), without LDS: one ballot per bit of the counts.  `total` is wave-uniform.
RT_DEV int pt_prefix(int x, int &total) {
    int before = 0;
    total = 0;
    for (int b = 0; b < 6; b++) {
        if (!pt_ballot((x >> b) != 0)) break;      // no lane has a bit at or above b (sparse queues: one or two rounds)
        const unsigned long long m = pt_ballot((x >> b) & 1);
        before += (int)pt_rank_below(m) << b;
        total += __popcll(m) << b;
    }
    return before;
}

// Hands paths to the lanes that want one.  The wave reads 64 bitmap words at once (lane i: word cursor + i), then the words
// claim for themselves — one LDS atomic instruction for all of them — and the claimed bits are dealt to the wanting lanes by rank, so a wave that
// wants 64 paths from a dense queue gets the two words of one 8x8 sub-tile (coherent rays) for two LDS atomics.  A word that
// holds more paths than are wanted keeps its upper bits and the cursor stays on it, so the next request starts there (no
// path is passed over).  Returns the local path index or PT_NONE.
// from_start: every request sweeps from word 0 — the paths at the front of the workgroup's list (its most expensive sub-tiles after a
// re-deal) are always served first, so the longest serial chains (a pixel's samples are serial) never wait behind cheap work.
RT_DEV uint32_t pt_pop(uint32_t *bm, int *cnt, const uint32_t nw, uint32_t &cursor, bool want, bool from_start = false) {
    const uint32_t lane = threadIdx.x & 63u;
    if (from_start) cursor = 0u;
    const unsigned long long wantmask = pt_ballot(want);
    const int need = __popcll(wantmask);
    const int my_rank = (int)pt_rank_below(wantmask);
    uint32_t got = PT_NONE;
    int have = 0;                                                 // wave-uniform, like everything below that is not per word (= per lane) or `got`
    for (uint32_t swept = 0; swept < nw && have < need; swept += 64u) {
        uint32_t w = cursor + lane;
        bool valid = true;
        if (nw >= 64u) { if (w >= nw) w -= nw; }
        else { valid = lane < nw; w = w % nw; }
        const uint32_t v = valid ? bm[w] : 0u;
        // Every word claims for itself, all in ONE LDS atomic of the wave: word i may take what the words before it leave of the request
        // (a prefix sum of the words' bit counts).  A claim can come back short (another wave was faster); the sweep then goes on.
        const int pc = __popc(v);
        int in_sight;
        const int before = pt_prefix(pc, in_sight);
        uint32_t next_cursor = cursor + 64u;
        if (in_sight) {
            const int room = need - have - before;
            uint32_t take = 0u;
            if (pc > 0 && room > 0) {
                take = v;
                if (room < pc) {                                  // the one word that is cut: its lowest `room` set bits
                    uint32_t rest = v;
                    for (int n = room; n > 0; n--) rest &= rest - 1u;
                    take = v & ~rest;
                }
            }
            uint32_t old = 0u;
            if (take) old = atomicAnd(&bm[w], ~take) & take;     // the bits this word really gave
            int claimed;
            const int first = pt_prefix(__popc(old), claimed);   // this word's paths go to the wanting lanes of ranks have + first, ...
            for (unsigned long long cm = pt_ballot(old != 0u); cm; cm &= cm - 1ull) {
                const int j = __ffsll((long long)cm) - 1;
                const uint32_t oj = (uint32_t)__builtin_amdgcn_readlane((int)old, j), wj = (uint32_t)__builtin_amdgcn_readlane((int)w, j);
                const int fj = have + __builtin_amdgcn_readlane(first, j);
                if (want && my_rank >= fj && my_rank < fj + __popc(oj)) {
                    uint32_t bits = oj;
                    for (int k = my_rank - fj; k > 0; k--) bits &= bits - 1u;
                    got = wj * 32u + (uint32_t)__ffs((int)bits) - 1u;
                }
            }
            have += claimed;
            const unsigned long long tm = pt_ballot(take != 0u);
            if (tm && have >= need) {                             // done: the next request starts at the last word touched if it kept paths, else behind it
                const int jl = 63 - __clzll((long long)tm);
                const uint32_t wl = (uint32_t)__builtin_amdgcn_readlane((int)w, jl), left = (uint32_t)__builtin_amdgcn_readlane((int)(v & ~take), jl);
                next_cursor = left ? wl : wl + 1u;
            }
        }
        cursor = (uint32_t)__builtin_amdgcn_readfirstlane((int)next_cursor);
        if (cursor >= nw) cursor %= nw;
    }
    if (have && lane == 0) atomicSub(cnt, have);
    return got;
}

// Sets the bit of path l in queue q for the lanes with `doit` (wave-uniform call).
template <class SH> RT_DEV void pt_push(SH &sh, int q, uint32_t l, bool doit) {
    if (doit) atomicOr(&sh.need[q][l >> 5], 1u << (l & 31u));
    const unsigned long long m = pt_ballot(doit);
    if (m && (threadIdx.x & 63u) == 0) atomicAdd(&sh.cnt[q], (int)__popcll(m));
}

// One of the two walks of path l is done (its results are in HBM, ordered before this call by the caller's release fence):
// clear its pending bit; whoever clears the last one hands the path to the shaders.  Wave-uniform call.
template <class SH> RT_DEV void pt_complete(SH &sh, uint32_t l, uint32_t bit, bool doit) {
    bool ready = false;
    if (doit) {
        const uint32_t shift = (l & 15u) * 2u;
        const uint32_t old = atomicAnd(&sh.pending[l >> 4], ~(bit << shift));
        ready = ((old >> shift) & 3u) == bit;
    }
    pt_push(sh, PT_Q_SHADE, l, ready);
}

// ---- closest-hit walker ------------------------------------------------------------------------------------------------
// The traversal loop of rt_wavefront.h (while-while, near-first, tie -> lowest figure index) fed from the need_trace bitmap.
// A finished lane keeps its path index in `fin` until the next refill point, where the wave orders its record stores before
// the LDS hand-off with one workgroup-scope release.
template <bool COUNT>
RT_DEV void pt_trace_stint(const SceneView &S, const WfView &W, PtShared &sh, const PtParams &P, PtWave &wv, uint32_t (*stack)[64],
                           const int shade_thr, uint32_t &n_queries, unsigned long long &n_nodes, unsigned long long &n_tris, PtProf &prof) {
    const int lane = threadIdx.x & 63;
    bool active = false, refill_ok = true, ending = false; // ending: the walk is over, its record is written at the next hand-off test
    uint32_t l = 0, slot = 0, cur = 0, hit = WF_MISS, fin = PT_NONE;
    int sp = 0;
    uint32_t pend = RT_EMPTY_LEAF, pend2 = RT_EMPTY_LEAF, pend3 = RT_EMPTY_LEAF; (void)pend; (void)pend2; (void)pend3; // PT_POSTPONE: the leaves this lane has met and not yet tested (pend first)
    uint32_t steps = 0;   // node steps + triangle tests of the lane's current walk: the cost measure of the re-deal (PT_COST_*)
    F3 o = f3(0.f, 0.f, 0.f), d = f3(0.f, 0.f, 1.f);
    PtRay ray = PT_RAY_IDLE; // idle lanes: never used
    float best_t = RT_T_MAX, best_u = 0.f, best_v = 0.f;
    // Boxes are pruned, and farther hits dropped, only beyond cull_t = best_t + the look-behind of rt_exact.h: the runner-up of the
    // best hit must be SEEN, whatever tree the walk uses, to decide at the end of the walk whether the exact walk is needed.
    float cull_t = RT_T_MAX, t2 = 2.f * RT_T_MAX, h_ray = 0.f; // h_ray: absolute part of the look-behind (pt_look_behind)
    auto store_hit = [&]() { // the gate (pt_shade_item) decides with the runner-up's t whether this hit needs the exact walk
        wf_rec(W, slot)[2] = make_float4(best_t, best_u, best_v, __uint_as_float(S.exact_boxes && hit != WF_MISS ? hit | pt_gap_code(best_t, t2) : hit));
        if (P.group_cost) atomicAdd(&sh.cost[l >> pt_gshift(sh)], steps);
    };
    PtLap<COUNT> clk;
    for (;;) {
        // Walks that ended since the last pass write their records here, together: in the loops below a lane only marks itself, so the
        // code of an ending (gap code, record store, cost counter) runs once per pass and not in every step in which some lane ends.
        if (pt_ballot(ending)) {
            if (ending) { store_hit(); fin = l; ending = false; }
        }
        const unsigned long long idle = pt_ballot(!active);
        if (idle && (__popcll(idle) >= (P.refill & 0xFFFF) || idle == ~0ull)) {
            // Hand-off point.  Finished lanes are published here and not the moment they finish: the release (a wait for the
            // wave's outstanding record stores) is paid once per refill, when the stores have long landed, not once per walk.
            PtLap<COUNT> sub;
            if (COUNT) prof.refills[0]++;
            if (pt_ballot(fin != PT_NONE)) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                pt_complete(sh, fin, PT_BIT_T, fin != PT_NONE);
                fin = PT_NONE;
            }
            sub.lap(prof.t_sub[0][0]);
            if (!refill_ok) {}
            else if (pt_count(&sh.cnt[PT_Q_SHADE]) >= shade_thr) refill_ok = false;      // shaders are behind: drain, then help them
            else if (pt_count(&sh.cnt[PT_Q_TRACE]) > 0) {
                const uint32_t got = pt_pop(sh.need[PT_Q_TRACE], &sh.cnt[PT_Q_TRACE], wv.nw, wv.cur[PT_Q_TRACE], !active, wv.front_first);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                sub.lap(prof.t_sub[0][1]);
                n_queries += __popcll(pt_ballot(got != PT_NONE));
                if (got != PT_NONE) {
                    l = got; slot = pt_slot(sh, l);
                    const float4 *r = wf_rec(W, slot);
                    float4 q0 = r[0], q1 = r[1];
                    o = f3(q0.x, q0.y, q0.z); d = f3(q0.w, q1.x, q1.y);
                    ray = pt_make_ray(S, o, d);
                    h_ray = S.exact_boxes ? pt_look_behind_abs(d, S.box_c2x) : 0.f;
                    steps = 0;
                    cur = 0; sp = 0; pend = RT_EMPTY_LEAF; pend2 = RT_EMPTY_LEAF; pend3 = RT_EMPTY_LEAF; hit = WF_MISS; best_t = RT_T_MAX; cull_t = RT_T_MAX; t2 = 2.f * RT_T_MAX; best_u = 0.f; best_v = 0.f;
                    active = true;
                }
                if (COUNT) { asm volatile("" : "+v"(cull_t)); sub.lap(prof.t_sub[0][2]); }
            }
        }
        const unsigned long long m_active = pt_ballot(active);
        clk.lap(prof.t_part[0][0]);
        if (!m_active) break;
        const int lb = pt_leaf_batch(P.leaf_batch, m_active);
        for (;;) { // phase 1: inner nodes
#if PT_POSTPONE
            // A lane that meets a leaf keeps it for the next leaf phase and walks on with what its stack holds (a second leaf stops it):
            // more lanes stay in the node loop, and more of them bring a leaf to each leaf phase.
            if (active && (cur & RT_LEAF_BIT) && PT_LAST_SLOT == RT_EMPTY_LEAF && cur != PT_DRAINED) {
                if (pend == RT_EMPTY_LEAF) pend = cur; else if (pend2 == RT_EMPTY_LEAF) pend2 = cur; else pend3 = cur; // (an empty leaf leaves the slot as it was)
                cur = sp == 0 ? PT_DRAINED : stack[--sp][lane];
                if (cur == PT_DRAINED && pend == RT_EMPTY_LEAF) { active = false; ending = true; } // an empty leaf was all that was left
            }
#endif
            const bool inner = active && !(cur & RT_LEAF_BIT);
            if (!pt_ballot(inner) || __popcll(pt_ballot(active && (cur & RT_LEAF_BIT))) >= lb) break;
            if (COUNT) { prof.trace_iters++; prof.trace_lane_iters += __popcll(pt_ballot(inner)); }
            if (inner) {
                if (COUNT) n_nodes++;
                steps += PT_COST_TRACE_NODE;
#if PT_QUANT_NODES
                const int went = pt_wide_step_nearest(S.nodes4, ray, cull_t, stack, lane, sp, P8_STACK, cur);
                if (went == PT_WIDE_NONE) {
#if PT_POSTPONE
                    if (sp != 0) cur = stack[--sp][lane];
                    else if (pend != RT_EMPTY_LEAF) cur = PT_DRAINED;
                    else { active = false; ending = true; }
#else
                    if (sp == 0) { active = false; ending = true; }
                    else cur = stack[--sp][lane];
#endif
                } else if (went == PT_WIDE_FULL) {
                    // The column is full (a walk holds up to three entries per level of a tree of up to P8_STACK / 2 levels; this takes a
                    // ray that grazes many boxes: triangle soups).  The walk ends here and says so — no hit has a negative t — and the
                    // exact role walks the query with a stack of its own (pt_exact_batch).
                    best_t = PT_T_OVERFLOW; best_u = 0.f; best_v = 0.f; hit = 0u; t2 = PT_T_OVERFLOW;
                    active = false; ending = true; pend = RT_EMPTY_LEAF; pend2 = RT_EMPTY_LEAF; pend3 = RT_EMPTY_LEAF;
                }
#else
                float n0, n1;
                const float4 *q = reinterpret_cast<const float4 *>(S.nodes + cur);
                float4 lo0 = q[0], hi0 = q[1], lo1 = q[2], hi1 = q[3];
                bool h0 = slab_test(lo0, hi0, ray, cull_t, n0);
                bool h1 = slab_test(lo1, hi1, ray, cull_t, n1);
                uint32_t c0 = __float_as_uint(lo0.w), c1 = __float_as_uint(lo1.w);
                if (h0 & h1) {
                    bool swap = n1 < n0;
                    stack[sp++][lane] = swap ? c0 : c1;
                    cur = swap ? c1 : c0;
                } else if (h0) cur = c0;
                else if (h1) cur = c1;
                else if (sp == 0) {
                    store_hit();
                    active = false; fin = l;
                } else cur = stack[--sp][lane];
#endif
            }
        }
        clk.lap(prof.t_part[0][1]);
#if PT_POSTPONE
        const bool at_leaf = active && pend != RT_EMPTY_LEAF;
        const uint32_t leaf = pend;
        uint32_t more = pend2, more2 = pend3;
#else
        const bool at_leaf = active && (cur & RT_LEAF_BIT);
        const uint32_t leaf = cur;
        uint32_t more = RT_EMPTY_LEAF, more2 = RT_EMPTY_LEAF;
#endif
        if (COUNT) { prof.leaf_iters[0]++; prof.leaf_lane_iters[0] += __popcll(pt_ballot(at_leaf)); }
        if (at_leaf) { // phase 2: leaves
            if (leaf != RT_EMPTY_LEAF) {
                uint32_t i = leaf & ~RT_LEAF_BIT;
                for (;;) {
                    TriIsect T = load_isect(S.tri_walk + i);
                    if (COUNT) n_tris++;
                    steps += PT_COST_TRACE_TRI;
                    float t, u, v; bool inside;
                    const uint32_t fi = T.pad >> 1; // index in the figure order
                    if (tri_test_closer(T, o, d, cull_t, t, u, v, inside)) {
                        const uint32_t best_i = hit & WF_INDEX_MASK;
                        if (t < best_t || (t == best_t && fi < best_i)) { // reference tie rule: smallest t, equal t -> lowest figure index
                            t2 = fminf(t2, best_t);
                            best_t = t; best_u = u; best_v = v; hit = fi | (inside ? WF_INSIDE_BIT : 0u);
                            cull_t = t + fmaxf(S.cull_k * t, h_ray);
                        } else t2 = fminf(t2, t);
                    }
                    if (!(T.pad & 1u)) i++;
                    else if (more == RT_EMPTY_LEAF) break;
                    else { i = more & ~RT_LEAF_BIT; more = more2; more2 = RT_EMPTY_LEAF; } // the lane's next leaf
                }
            }
#if PT_POSTPONE
            pend = RT_EMPTY_LEAF; pend2 = RT_EMPTY_LEAF; pend3 = RT_EMPTY_LEAF;
            if (cur == PT_DRAINED) { active = false; ending = true; } // a lane that stopped at a second leaf keeps that for the next round
#else
            if (sp == 0) {
                store_hit();
                active = false; fin = l;
            } else cur = stack[--sp][lane];
#endif
        }
        clk.lap(prof.t_part[0][2]);
    }
}

// ---- light-sum walker (wf_light_loop_lean of rt_wavefront.h fed from the need_light bitmap) ----------------------------------
template <bool COUNT>
RT_DEV void pt_light_stint(const SceneView &S, const WfView &W, PtShared &sh, const PtParams &P, PtWave &wv, uint32_t (*stack)[64],
                           const int shade_thr, uint32_t &n_queries, unsigned long long &n_nodes, unsigned long long &n_tris, PtProf &prof) {
    const int lane = threadIdx.x & 63;
    bool active = false, overflow = false, refill_ok = true, ending = false; // ending: see pt_trace_stint
    uint32_t l = 0, slot = 0, cur = 0, fin = PT_NONE; // fin: the lane's finished, unpublished path; bit 31 = it needs the exact role instead
    int sp = 0, k = 0;
    uint32_t pend = RT_EMPTY_LEAF, pend2 = RT_EMPTY_LEAF, pend3 = RT_EMPTY_LEAF; (void)pend; (void)pend2; (void)pend3; // PT_POSTPONE: the leaves this lane has met and not yet tested (pend first)
    uint32_t steps = 0;
    F3 o = f3(0.f, 0.f, 0.f), d = f3(0.f, 0.f, 1.f);
    PtRay ray = PT_RAY_IDLE; // idle lanes: never used
    auto finish = [&]() {
        active = false;
        if (P.group_cost) atomicAdd(&sh.cost[l >> pt_gshift(sh)], steps);
        if (overflow) { fin = l | 0x80000000u; return; }
        float v = 0.f;
        if (k == 1) v = __uint_as_float(stack[P8_STACK - 2][lane]);
        else if (k == 2) v = __uint_as_float(stack[P8_STACK - 2][lane]) + __uint_as_float(stack[P8_STACK - 4][lane]);
        else if (k > 2) { // the reference's association of the additions, see wf_light_loop_lean
            const uint32_t nl = S.n_lights;
            for (int j = 1; j < k; j++) {
                uint32_t a0 = stack[P8_STACK - 1 - 2 * (j - 1)][lane], b0 = stack[P8_STACK - 1 - 2 * j][lane];
                uint32_t len = b0 - a0;
                uint32_t lv = 31u - (uint32_t)__clz((int)len);
                uint16_t m0 = S.light_sep[(size_t)lv * nl + a0], m1 = S.light_sep[(size_t)lv * nl + (b0 - (1u << lv))];
                stack[j - 1][lane] = m0 < m1 ? m0 : m1;
            }
            for (int n = k; n > 1; n--) {
                int best = 1;
                uint32_t bd = stack[0][lane];
                for (int i = 2; i < n; i++) { uint32_t di = stack[i - 1][lane]; if (di > bd) { bd = di; best = i; } }
                float merged = __uint_as_float(stack[P8_STACK - 2 - 2 * (best - 1)][lane]) + __uint_as_float(stack[P8_STACK - 2 - 2 * best][lane]);
                stack[P8_STACK - 2 - 2 * (best - 1)][lane] = __float_as_uint(merged);
                for (int i = best; i < n - 1; i++) {
                    stack[P8_STACK - 2 - 2 * i][lane] = stack[P8_STACK - 2 - 2 * (i + 1)][lane];
                    stack[i - 1][lane] = stack[i][lane];
                }
            }
            v = __uint_as_float(stack[P8_STACK - 2][lane]);
        }
        int depth = (int)(__float_as_uint(reinterpret_cast<const float *>(wf_rec(W, slot) + 3)[3]) & 15u);
        float *pdf = reinterpret_cast<float *>(wf_entry(W, slot, depth)) + 3;
        *pdf = *pdf + v / S.n_lights_f;                                  // distributions.h:123,273
        fin = l;
    };
    PtLap<COUNT> clk;
    for (;;) {
        if (pt_ballot(ending)) { // the sums of the walks that ended since the last pass (pt_trace_stint)
            if (ending) { finish(); ending = false; }
        }
        const unsigned long long idle = pt_ballot(!active);
        if (idle && (__popcll(idle) >= (P.refill >> 16) || idle == ~0ull)) {
            if (pt_ballot(fin != PT_NONE)) { // hand-off point, see pt_trace_stint
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                const bool slow = (fin >> 31) != 0u && fin != PT_NONE;
                pt_complete(sh, fin, PT_BIT_L, fin != PT_NONE && !slow);
                pt_push(sh, PT_Q_XLIGHT, fin & 0x7FFFFFFFu, slow);
                fin = PT_NONE;
            }
            if (!refill_ok) {}
            else if (pt_count(&sh.cnt[PT_Q_SHADE]) >= shade_thr) refill_ok = false;
            else if (pt_count(&sh.cnt[PT_Q_LIGHT]) > 0) {
                const uint32_t got = pt_pop(sh.need[PT_Q_LIGHT], &sh.cnt[PT_Q_LIGHT], wv.nw, wv.cur[PT_Q_LIGHT], !active, wv.front_first);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                n_queries += __popcll(pt_ballot(got != PT_NONE));
                if (got != PT_NONE) {
                    l = got; slot = pt_slot(sh, l);
                    const float4 *r = wf_rec(W, slot);
                    float4 q0 = r[0], q1 = r[1];
                    o = f3(q0.x, q0.y, q0.z); d = f3(q0.w, q1.x, q1.y);
                    ray = pt_make_ray(S, o, d);
                    cur = 0; sp = 0; pend = RT_EMPTY_LEAF; pend2 = RT_EMPTY_LEAF; pend3 = RT_EMPTY_LEAF; k = 0; overflow = false; steps = 0;
                    active = true;
                }
            }
        }
        const unsigned long long m_active = pt_ballot(active);
        clk.lap(prof.t_part[1][0]);
        if (!m_active) break;
        const int lb = pt_leaf_batch(P.leaf_batch, m_active);
        for (;;) { // phase 1: inner nodes
#if PT_POSTPONE
            if (active && (cur & RT_LEAF_BIT) && PT_LAST_SLOT == RT_EMPTY_LEAF && cur != PT_DRAINED) { // the leaf waits for the next leaf phase (pt_trace_stint)
                if (pend == RT_EMPTY_LEAF) pend = cur; else if (pend2 == RT_EMPTY_LEAF) pend2 = cur; else pend3 = cur;
                cur = sp == 0 ? PT_DRAINED : stack[--sp][lane];
                if (cur == PT_DRAINED && pend == RT_EMPTY_LEAF) { active = false; ending = true; }
            }
#endif
            const bool inner = active && !(cur & RT_LEAF_BIT);
            if (!pt_ballot(inner) || __popcll(pt_ballot(active && (cur & RT_LEAF_BIT))) >= lb) break;
            if (COUNT) { prof.light_iters++; prof.light_lane_iters += __popcll(pt_ballot(inner)); }
            if (inner) {
                if (COUNT) n_nodes++;
                steps += PT_COST_LIGHT_NODE;
#if PT_QUANT_NODES
                const int went = pt_wide_step_all(S.light_walk_nodes4, ray, stack, lane, sp, P8_STACK - 2 * k - 1, cur); // the hits sit at the column's top
                if (went == PT_WIDE_NONE) {
#if PT_POSTPONE
                    if (sp != 0) cur = stack[--sp][lane];
                    else if (pend != RT_EMPTY_LEAF) cur = PT_DRAINED;
                    else { active = false; ending = true; }
#else
                    if (sp == 0) { active = false; ending = true; }
                    else cur = stack[--sp][lane];
#endif
                } else if (went == PT_WIDE_FULL) { overflow = true; pend = RT_EMPTY_LEAF; pend2 = RT_EMPTY_LEAF; pend3 = RT_EMPTY_LEAF; { active = false; ending = true; } } // no room beside the hits: the slow role sums this query
#else
                float n0, n1;
                const float4 *q = reinterpret_cast<const float4 *>(S.light_walk_nodes + cur);
                float4 lo0 = q[0], hi0 = q[1], lo1 = q[2], hi1 = q[3];
                bool h0 = slab_test(lo0, hi0, ray, RT_T_MAX, n0);
                bool h1 = slab_test(lo1, hi1, ray, RT_T_MAX, n1);
                uint32_t c0 = __float_as_uint(lo0.w), c1 = __float_as_uint(lo1.w);
                if (h0 & h1) { stack[sp++][lane] = c1; cur = c0; if (sp + 2 * k >= P8_STACK) overflow = true; }
                else if (h0) cur = c0;
                else if (h1) cur = c1;
                else if (sp == 0) { active = false; ending = true; }
                else cur = stack[--sp][lane];
#endif
            }
        }
        clk.lap(prof.t_part[1][1]);
#if PT_POSTPONE
        const bool at_leaf = active && pend != RT_EMPTY_LEAF;
        const uint32_t leaf = pend;
        uint32_t more = pend2, more2 = pend3;
#else
        const bool at_leaf = active && (cur & RT_LEAF_BIT);
        const uint32_t leaf = cur;
        uint32_t more = RT_EMPTY_LEAF, more2 = RT_EMPTY_LEAF;
#endif
        if (COUNT) { prof.leaf_iters[1]++; prof.leaf_lane_iters[1] += __popcll(pt_ballot(at_leaf)); }
        if (at_leaf) { // phase 2: leaves
            if (leaf != RT_EMPTY_LEAF) {
                uint32_t i = leaf & ~RT_LEAF_BIT;
                // The loop only tests; what a hit needs beyond the test (the rest of the light's record, the pdf term, the robustness test, the
                // sorted insertion) waits until after the loop — a lane rarely hits twice in one leaf phase, and the long hit code then runs once
                // per phase instead of once per tested light.
                bool held = false; uint32_t h_i = 0u, h_li = 0u; float h_t = 0.f, h_u = 0.f, h_v = 0.f; bool h_in = false;
                auto take = [&]() { // the held hit joins the lane's hits
                    bool robust;
                    const float term = pt_light_pdf_hit(S, S.lights_walk + h_i, o, d, h_t, h_u, h_v, h_in, robust);
                    if (COUNT && term != 0.f) prof.light_hits++;
                    if (term != 0.f) { // (a hit whose term is exactly 0 adds nothing, like a miss)
                        if (!robust || k >= WF_MAX_LIGHT_HITS || sp + 2 * k + 2 >= P8_STACK) overflow = true;
                        else { // kept sorted by light index (this tree's leaf order is not the light order): hit j at words P8_STACK-1-2j (index), -2-2j (term)
                            int j = k;
                            while (j > 0 && stack[P8_STACK - 1 - 2 * (j - 1)][lane] > h_li) {
                                stack[P8_STACK - 1 - 2 * j][lane] = stack[P8_STACK - 1 - 2 * (j - 1)][lane];
                                stack[P8_STACK - 2 - 2 * j][lane] = stack[P8_STACK - 2 - 2 * (j - 1)][lane];
                                j--;
                            }
                            stack[P8_STACK - 1 - 2 * j][lane] = h_li; stack[P8_STACK - 2 - 2 * j][lane] = __float_as_uint(term); k++;
                        }
                    }
                    held = false;
                };
                for (;;) {
                    bool last, inside; uint32_t li; float t, u, v;
                    if (COUNT) { n_tris++; prof.light_tests++; }
                    steps += PT_COST_LIGHT_TEST;
                    if (pt_light_test(S.lights_walk + i, o, d, last, li, t, u, v, inside)) {
                        if (held) take(); // a second hit in this phase
                        held = true; h_i = i; h_li = li; h_t = t; h_u = u; h_v = v; h_in = inside;
                    }
                    if (!last) i++;
                    else if (more == RT_EMPTY_LEAF) break;
                    else { i = more & ~RT_LEAF_BIT; more = more2; more2 = RT_EMPTY_LEAF; } // the lane's next leaf
                }
                if (held) take();
            }
#if PT_POSTPONE
            pend = RT_EMPTY_LEAF; pend2 = RT_EMPTY_LEAF; pend3 = RT_EMPTY_LEAF;
            if (cur == PT_DRAINED) { active = false; ending = true; }
#else
            if (sp == 0) { active = false; ending = true; }
            else cur = stack[--sp][lane];
#endif
        }
        clk.lap(prof.t_part[1][2]);
    }
}

// ---- the shader role: wf_shade_item / pt_shade_item of rt_wavefront.h laid out for a 96-VGPR budget -----------------------------------
// Same arithmetic, same order of random draws, same record writes.  What differs is where values wait: the shading code is a chain of
// sections (finish the pending bounce | hit attributes and textures | Mix::sample | BRDF | Mix::pdf terms | path epilogue), each of
// which needs 50-75 VGPRs on its own, and only what the *current* section works on stays in registers.  Everything else that a later
// section needs — the incoming direction, the random engine, base colour x texture colour, the metallic product, a path's tail —
// is parked in this lane's column of the wave's LDS stack area (idle while the wave shades; volatile accesses, so the compiler neither
// forwards a parked value through a register nor moves other memory operations across a park / unpark).
#define PK_TAIL 0                     // 3 words: value the innermost call returns (paths that end)
#define PK_LEVELS 3                   // bounces below which it returns
#define PK_RNG 4                      // engine state, saved normal (has_saved travels in the packed word)
#define PK_D 6                        // incoming direction (3)
#define PK_BC 9                       // base_color * texture colour (3), metallic * baseMetallic
#define PK_WORDS 13
static_assert(PK_WORDS <= P8_STACK, "the shader parks its values in the lane's stack column");
typedef __attribute__((address_space(3))) volatile uint32_t *PtLdsWord; // an LDS pointer that stays one (32-bit base + immediate offsets)
struct PtPark {
    PtLdsWord p;                      // this lane's column: word i at p[64 * i]
    RT_DEV void put(int i, float v) const { p[64 * i] = __float_as_uint(v); }
    RT_DEV void putu(int i, uint32_t v) const { p[64 * i] = v; }
    RT_DEV float get(int i) const { return __uint_as_float(p[64 * i]); }
    RT_DEV uint32_t getu(int i) const { return p[64 * i]; }
    RT_DEV void put3(int i, F3 v) const { put(i, v.x); put(i + 1, v.y); put(i + 2, v.z); }
    RT_DEV F3 get3(int i) const { const float x = get(i), y = get(i + 1), z = get(i + 2); return f3(x, y, z); }
    RT_DEV Rng rng(uint32_t packed) const { Rng g; g.x = getu(PK_RNG); g.saved = get(PK_RNG + 1); g.has_saved = (packed & 16u) != 0; return g; }
    RT_DEV void keep(const Rng &g, uint32_t &packed) const { putu(PK_RNG, g.x); put(PK_RNG + 1, g.saved); packed = g.has_saved ? (packed | 16u) : (packed & ~16u); }
    RT_DEV void end(F3 tail, int levels) const { put3(PK_TAIL, tail); putu(PK_LEVELS, (uint32_t)levels); }
};

template <int FEAT>
RT_DEV int pt_shade_lean(const SceneView &S, const RenderView &R, const WfView &W, const uint32_t slot, const PtPark pk, bool &discarded) {
    float4 *r = wf_rec(W, slot);
    uint32_t packed = __float_as_uint(reinterpret_cast<const float *>(r + 3)[3]);
    int depth = (int)(packed & 15u);
    const float4 q0 = r[0], q1 = r[1], q2 = r[2];
    const uint32_t hit = __float_as_uint(q2.w);
    // Everything whose address is known once the record is in goes out together — the figure's box (the gate), its plane normal, the head of
    // its TriShade record (texture coordinates, material), the pending bounce's entry — so that ONE memory latency covers what would otherwise
    // be four dependent round trips (gate -> entry -> normal -> attributes).  Lanes without a hit read figure 0, lanes without a pending
    // bounce read their level's entry anyway: the values are simply not used.
    const uint32_t fig = hit != WF_MISS ? (hit & WF_INDEX_MASK) : 0u;
    float4 blo = make_float4(0.f, 0.f, 0.f, 0.f), bhi = blo;
    if (S.exact_boxes) { const float4 *bx = reinterpret_cast<const float4 *>(S.tri_box) + 2 * (size_t)fig; blo = bx[0]; bhi = bx[1]; }
    const F3 tri_n = load_tri_normal(S, fig);
    const TriShadeHead head = load_shade_head(S, fig);
    float4 pe0, pe1;
    { const float4 *e = wf_entry(W, slot, depth); pe0 = e[0]; pe1 = e[1]; }
    // the exactness gate (pt_shade_item): a hit that does not stand as the reference's answer goes to the exact walk first, untouched
    if (q2.x == PT_T_OVERFLOW && !(packed & WF_VERIFIED_BIT)) return PT_SHADE_EXACT; // the walk ran out of stack (pt_trace_stint)
    if (S.exact_boxes == 1u && hit != WF_MISS && !(packed & WF_VERIFIED_BIT) &&
        !pt_hit_stands(f3(blo.x, blo.y, blo.z), f3(bhi.x, bhi.y, bhi.z), f3(q0.x, q0.y, q0.z), f3(q0.w, q1.x, q1.y), q2.x, pt_gap_floor(hit, q2.x), S.box_c2, S.box_c2x, S.cull_k))
        return PT_SHADE_EXACT;
    pk.put(PK_D, q0.w); pk.put(PK_D + 1, q1.x); pk.put(PK_D + 2, q1.y);
    pk.putu(PK_RNG, __float_as_uint(q1.z)); pk.put(PK_RNG + 1, q1.w);
    bool ended = false;
    if (packed & WF_PENDING_BIT) {
        // The bounce at `depth` sampled the ray that was just traced; its pdf is complete now (Mix::pdf, distributions.h:268-278).
        float4 *e = wf_entry(W, slot, depth);
        const float4 e0 = pe0, e1 = pe1;
        const float pdf = e0.w / S.n_components_f;                                  // :278
        const float k = (float)(1. / (double)pdf * fabs((double)e1.w));             // scene.cpp:159
        F3 mult = k * f3(e1.x, e1.y, e1.z);
        const bool clamp = mult.x > 6.f || mult.y > 6.f || mult.z > 6.f || mult.x != mult.x || mult.y != mult.y || mult.z != mult.z;
        if (clamp || depth + 1 >= R.ray_depth) {
            // clamp hack (scene.cpp:161-163): the path returns the emission and the speculative hit is dropped; at the
            // last level the inner call returns 0, i.e. emission + mult * 0 evaluated literally.
            discarded = true; ended = true;
            if (clamp) pk.end(f3(e0.x, e0.y, e0.z), depth);
            else { e[1] = make_float4(mult.x, mult.y, mult.z, e1.w); pk.end(f3(0.f, 0.f, 0.f), depth + 1); }
        } else {
            bool survives = true;
            if (R.rr_depth > 0 && depth + 1 >= R.rr_depth) { // Russian roulette (throughput mode only), see wf_shade_item
                Rng rng = pk.rng(packed);
                const float q = wf_roulette_q(W, slot, depth, mult);
                survives = rng_u01(rng) < q;
                mult = (1.f / q) * mult;
                pk.keep(rng, packed);
            }
            e[1] = make_float4(mult.x, mult.y, mult.z, e1.w);
            if (survives) depth++;
            else { discarded = true; ended = true; pk.end(f3(0.f, 0.f, 0.f), depth + 1); }
        }
    }
    if (!ended) {
        HitRec h;
        h.idx = (int)(hit & WF_INDEX_MASK); h.inside = (hit & WF_INSIDE_BIT) != 0; h.t = q2.x; h.u = q2.y; h.v = q2.z;
        if (hit == WF_MISS) { ended = true; pk.end(miss_color<(FEAT & WF_FEAT_ENV) != 0>(S, pk.get3(PK_D)), depth); }
        else if (S.last_level_emission_only && depth + 1 >= R.ray_depth) {
            // Deepest level: getColor returns its emission whatever Mix::sample / brdf / pdf produce (SceneView::last_level_emission_only);
            // only the random draws must still happen, in order (distributions.h:257, then 3 normals | u1,u2 | index,u,v).
            pk.end(emission_fetch(S, h), depth);
            Rng rng = pk.rng(packed);
            const int comp = (int)(rng_u01(rng) * S.n_components_f);
            if (comp == 0) { rng_n01(rng); rng_n01(rng); rng_n01(rng); }
            else if (comp == 2) { rng_u01(rng); rng_u01(rng); rng_u01(rng); }
            else { rng_u01(rng); rng_u01(rng); }
            pk.keep(rng, packed);
            ended = true;
        } else {
            const bool hw7 = (FEAT & WF_FEAT_HW7) && S.hw7;
            float4 *e = wf_entry(W, slot, depth);
            {   // the next ray's origin goes to the path's record at once (its final place): x + eps * geomNorma, scene.cpp:104
                const F3 ng = geom_normal(tri_n, h.inside);
                const F3 x = f3(q0.x, q0.y, q0.z) + h.t * pk.get3(PK_D);
                const F3 xo = x + 9.99999974737875163555e-05f * ng;
                r[0] = make_float4(xo.x, xo.y, xo.z, 0.f);                           // the next ray doubles as the light query
            }
            float alpha; F3 sn;
            {
                // hit attributes, material, textures (scene.cpp:99-149).  The emission goes to the level's entry at once; colour and
                // metallic shrink to the products the BRDF uses and wait in the park.
                F3 base_color; float base_metallic; Shaded sh;
                shade_fetch_attr(S, h, head, sh, base_color, base_metallic, hw7);
                e[0] = make_float4(sh.emission.x, sh.emission.y, sh.emission.z, 0.f);
                pk.put3(PK_BC, base_color * sh.color);
                pk.put(PK_BC + 3, sh.metallic * base_metallic);
                alpha = sh.alpha; sn = sh.sn;
            }
            F3 nd;
            {   // Mix::sample (distributions.h:256-265)
                Rng rng = pk.rng(packed);
                const int comp = (int)(rng_u01(rng) * S.n_components_f);         // :257
                if (comp == 0) nd = cosine_sample(rng, sn);
                else if (comp == 2) { const float4 xq = r[0]; nd = light_sample(S, rng, f3(xq.x, xq.y, xq.z)); }
                else nd = vndf_sample(rng, sn, pk.get3(PK_D), alpha);
                pk.keep(rng, packed);
            }
            // the record and entry addresses are formed again from the slot number after the sampling code (an opaque copy, so that the
            // two 64-bit pointers do not sit in registers across it)
            uint32_t slot_again = slot;
            int depth_again = depth;
            asm volatile("" : "+v"(slot_again), "+v"(depth_again));
            r = wf_rec(W, slot_again);
            e = wf_entry(W, slot_again, depth_again);
            F3 brdf;
            {
                const F3 d = pk.get3(PK_D), bc = pk.get3(PK_BC);
                const float metallic_eff = pk.get(PK_BC + 3);
                brdf = hw7 ? material_brdf_hw7(bc, metallic_eff, nd, neg(d), sn, alpha * alpha)
                           : material_brdf_pre(bc, metallic_eff, nd, neg(d), sn, alpha);
            }
            const float epsf = 9.99999974737875163555e-05f;
            if (brdf.x <= epsf && brdf.y <= epsf && brdf.z <= epsf) {                 // scene.cpp:154-156
                const float4 e0 = e[0];
                ended = true; pk.end(f3(e0.x, e0.y, e0.z), depth);
            } else {
                e[1] = make_float4(brdf.x, brdf.y, brdf.z, dot(nd, sn));
                asm volatile("" : "+v"(sn.x), "+v"(sn.y), "+v"(sn.z)); // the pdf terms start from the normal again: nothing derived from it for the sampling code (its rotation) waits in registers
                float pdf = 0.f;                                                       // distributions.h:268-276, first two terms
                pdf += cosine_pdf(sn, nd);
                pdf += vndf_pdf(sn, nd, pk.get3(PK_D), alpha);
                reinterpret_cast<float *>(e)[3] = pdf;
                reinterpret_cast<float *>(r)[3] = nd.x;
                r[1] = make_float4(nd.y, nd.z, __uint_as_float(pk.getu(PK_RNG)), pk.get(PK_RNG + 1));
                const uint32_t sample = (packed >> 6) & WF_SAMPLE_MASK;
                reinterpret_cast<float *>(r + 3)[3] = __uint_as_float(wf_pack(depth, (packed & 16u) != 0, sample, true));
                return WF_NEXT_TRACE | (S.n_lights ? WF_NEXT_LIGHT : 0);               // traced speculatively beside its own light-pdf sum
            }
        }
    }
    Rng rng = pk.rng(packed);
    return wf_finish_path(S, R, W, slot, (int)pk.getu(PK_LEVELS), pk.get3(PK_TAIL), rng, (packed >> 6) & WF_SAMPLE_MASK);
}

// ---- the exact role: one lane per query, the reference's own box arithmetic over the reference trees (rt_exact.h) -------------------
// At most PT_EXACT_BATCH queries of each kind per call; their node stacks live in the wave's (otherwise idle) LDS stack area, entry e of
// lane i at word e * PT_EXACT_BATCH + i — no scratch.  Rare (3e-5 of the queries), so the partly filled wave does not matter.
template <class SH>
RT_DEV void pt_exact_batch(const SceneView &S, const WfView &W, SH &sh, PtWave &wv, uint32_t *area, uint32_t &n_xlight, uint32_t &n_xtrace) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t *xstack = area + (lane & (PT_EXACT_BATCH - 1u));
    uint32_t got = pt_pop(sh.need[PT_Q_XLIGHT], &sh.cnt[PT_Q_XLIGHT], wv.nw, wv.cur[PT_Q_XLIGHT], lane < PT_EXACT_BATCH);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (got != PT_NONE) {
        const uint32_t slot = pt_slot(sh, got);
        const float4 *r = wf_rec(W, slot);
        float4 q0 = r[0], q1 = r[1];
        const F3 x = f3(q0.x, q0.y, q0.z), d = f3(q0.w, q1.x, q1.y);
        float v;
        if (S.exact_boxes) v = ref_light_pdf_sum<PT_EXACT_BATCH>(S, x, d, xstack);
        else { Counters c; c.closest = c.lightq = c.nodes = c.tris = 0; v = light_pdf_sum<false, PT_EXACT_BATCH>(S, x, d, xstack, c); }
        int depth = (int)(__float_as_uint(r[3].w) & 15u);
        float *pdf = reinterpret_cast<float *>(wf_entry(W, slot, depth)) + 3;
        *pdf = *pdf + v / S.n_lights_f;
    }
    n_xlight += __popcll(pt_ballot(got != PT_NONE));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    pt_complete(sh, got, PT_BIT_L, got != PT_NONE);
    got = pt_pop(sh.need[PT_Q_XTRACE], &sh.cnt[PT_Q_XTRACE], wv.nw, wv.cur[PT_Q_XTRACE], lane < PT_EXACT_BATCH);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (got != PT_NONE) {
        const uint32_t slot = pt_slot(sh, got);
        float4 *r = wf_rec(W, slot);
        float4 q0 = r[0], q1 = r[1];
        float bt, bu, bv; uint32_t hit;
        if (S.exact_boxes == 1u) ref_closest_hit<PT_EXACT_BATCH>(S, f3(q0.x, q0.y, q0.z), f3(q0.w, q1.x, q1.y), xstack, bt, bu, bv, hit);
        else { // no reference boxes to be exact about (a walk that ran out of stack): the padded float boxes of the two-box tree decide, as for every other hit
            Counters c; c.closest = c.lightq = c.nodes = c.tris = 0;
            const HitRec h = closest_hit<false, PT_EXACT_BATCH>(S, f3(q0.x, q0.y, q0.z), f3(q0.w, q1.x, q1.y), xstack, c);
            bt = h.t; bu = h.u; bv = h.v; hit = h.idx < 0 ? WF_MISS : ((uint32_t)h.idx | (h.inside ? WF_INSIDE_BIT : 0u));
        }
        r[2] = make_float4(bt, bu, bv, __uint_as_float(hit));
        float *pk = reinterpret_cast<float *>(r + 3) + 3;
        *pk = __uint_as_float(__float_as_uint(*pk) | WF_VERIFIED_BIT);
    }
    n_xtrace += __popcll(pt_ballot(got != PT_NONE));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    // A hit the gate sent here has both walks behind it: it goes back to the shaders.  A ray that came here instead of to the walkers (a
    // tripwire, pt_tripwire) still has its closest-hit bit pending, and its light sum may be under way: the last of the two hands it on.
    bool ready = false;
    if (got != PT_NONE) {
        const uint32_t shift = (got & 15u) * 2u;
        const uint32_t old = (atomicAnd(&sh.pending[got >> 4], ~(PT_BIT_T << shift)) >> shift) & 3u;
        ready = old == 0u || old == PT_BIT_T;
    }
    pt_push(sh, PT_Q_SHADE, got, ready);
}

// ---- the kernel -----------------------------------------------------------------------------------------------------------
// __launch_bounds__(256, 5): five waves per SIMD, i.e. a budget of 96 VGPRs.  Every role fits it without scratch; the scheduler's own
// state is wave-uniform and lives in SGPRs (pt_count / readfirstlane).
template <bool COUNT, int FEAT>
__global__ __launch_bounds__(P8_THREADS, P8_PER_CU) void pt_persistent_kernel(SceneView S, RenderView R, WfView W, PtParams P) {
    __shared__ PtShared sh;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    PtWave wv;
    wv.n_blocks = gridDim.x; wv.block = blockIdx.x;
    wv.front_first = P.front_first != 0u;
    const uint32_t first_group = P.group_ofs ? P.group_ofs[wv.block] : 0u;
    const uint32_t n_local_groups = P.group_ofs ? P.group_ofs[wv.block + 1u] - first_group
                                                : (P.n_groups > wv.block ? (P.n_groups - wv.block + wv.n_blocks - 1u) / wv.n_blocks : 0u);
    wv.n_local = n_local_groups << P.group_shift;
    wv.nw = (wv.n_local + 31u) >> 5;
    if (wv.n_local == 0u) return;
    if (P.debug && tid == 0) { P.debug[3 * blockIdx.x] = __builtin_amdgcn_s_memrealtime(); P.debug[3 * blockIdx.x + 2] = wv.n_local; }
    for (int q = 0; q < 5; q++) wv.cur[q] = (wave * 64u) % wv.nw;

    // ---- init: seed every pixel of this workgroup, first camera ray (wf_init_kernel of rt_wavefront.h) -----------------------
    for (uint32_t i = tid; i < wv.nw; i += P8_THREADS) { sh.need[0][i] = 0; sh.need[1][i] = 0; sh.need[2][i] = 0; sh.need[3][i] = 0; sh.need[4][i] = 0; }
    for (uint32_t i = tid; i < 2u * wv.nw; i += P8_THREADS) sh.pending[i] = 0;
    for (uint32_t i = tid; i < n_local_groups; i += P8_THREADS) { sh.groups[i] = P.group_ofs ? P.group_ids[first_group + i] : i * wv.n_blocks + wv.block; sh.cost[i] = 0; }
    if (tid < 16u) sh.cnt[tid] = tid == PT_GSHIFT ? (int)P.group_shift : 0;
    __syncthreads();
    for (uint32_t base = 0; base < wv.n_local; base += P8_THREADS) {
        const uint32_t l = base + tid;
        bool started = false;
        if (l < wv.n_local) {
            const uint32_t slot = pt_slot(sh, l), gslot = slot + W.slot_base;
            int x, y; bool inside; size_t out_index;
            wf_slot_to_pixel(R, gslot, x, y, inside, out_index);
            if (!inside) { // padding of a border tile in the compact shard layout
                if (R.shard_count > 1 && (R.streams <= 1 || gslot < R.n_pixslots)) {
                    if (R.out_rgb) { R.out_rgb[3 * out_index] = 0.f; R.out_rgb[3 * out_index + 1] = 0.f; R.out_rgb[3 * out_index + 2] = 0.f; }
                    if (R.out_rgb8) { R.out_rgb8[3 * out_index] = 0; R.out_rgb8[3 * out_index + 1] = 0; R.out_rgb8[3 * out_index + 2] = 0; }
                }
            } else if (P.resume) {
                // a later phase of the frame: the record holds the pixel sum, the random stream and the parked camera ray
                const uint32_t packed = __float_as_uint(reinterpret_cast<const float *>(wf_rec(W, slot) + 3)[3]);
                started = ((packed >> 6) & WF_SAMPLE_MASK) < (uint32_t)R.samples;
                if (started) atomicOr(&sh.pending[l >> 4], PT_BIT_T << ((l & 15u) * 2u));
            } else {
                Rng rng;
                rng_seed(rng, (uint32_t)(y * R.width + x) + (R.streams > 1 ? (gslot / R.n_pixslots) * R.seed_stride : 0u)); // sceneio.cpp:389-391
                if (R.sample_seeds) wf_sample_seed(R, rng, gslot, x, y, 0u);
                F3 o, d;
                wf_camera_ray(S, R, rng, x, y, o, d);
                float4 *r = wf_rec(W, slot);
                r[0] = make_float4(o.x, o.y, o.z, d.x);
                r[1] = make_float4(d.y, d.z, __uint_as_float(rng.x), rng.saved);
                r[2] = make_float4(0.f, 0.f, 0.f, 0.f);
                r[3] = make_float4(0.f, 0.f, 0.f, __uint_as_float(wf_pack(0, rng.has_saved, 0)));
                atomicOr(&sh.pending[l >> 4], PT_BIT_T << ((l & 15u) * 2u));
                started = true;
            }
        }
        const unsigned long long m = pt_ballot(started);
        if (m && lane == 0) atomicAdd(&sh.cnt[PT_N_LIVE], (int)__popcll(m));
        bool wire = false;
        if (S.n_tripwire_groups && started) {
            const float4 *nr = wf_rec(W, pt_slot(sh, l));
            const float4 n0 = nr[0], n1 = nr[1];
            wire = pt_tripwire(S, f3(n0.x, n0.y, n0.z), f3(n0.w, n1.x, n1.y));
        }
        pt_push(sh, PT_Q_TRACE, l, started && !wire);
        pt_push(sh, PT_Q_XTRACE, l, wire);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

    // ---- scheduler: every wave picks a role whenever it is idle ----------------------------------------------------------------
    uint32_t(*stack)[64] = sh.stack[wave];
    const int shade_thr = P.shade_thr0 + (int)wave * P.shade_thr_step;
    uint32_t n_closest = 0, n_light = 0, n_xtrace = 0, n_xlight = 0, n_discarded = 0; // per wave and launch: well below 2^32
    unsigned long long n_nodes = 0, n_tris = 0;
    uint32_t idle_spins = 0;
    int gave_up = 0; // 1: the launch ran into its deadline; 2: the workgroup waited in vain for a path to come back (a lost path: a bug)
    PtProf prof;
    unsigned long long t_mark = COUNT ? __builtin_amdgcn_s_memtime() : 0ull;
    auto lap = [&](unsigned long long &acc) { if (COUNT) { const unsigned long long t = __builtin_amdgcn_s_memtime(); acc += t - t_mark; t_mark = t; } };
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        if (__builtin_amdgcn_s_memrealtime() - t_start > P.deadline_ticks) { gave_up = 1; break; } // safety net: never hang the GPU; the host reports the error
        const int ns = pt_count(&sh.cnt[PT_Q_SHADE]), nt = pt_count(&sh.cnt[PT_Q_TRACE]), nl = pt_count(&sh.cnt[PT_Q_LIGHT]);
        const int nx = pt_count(&sh.cnt[PT_Q_XLIGHT]) + pt_count(&sh.cnt[PT_Q_XTRACE]);
        if (nx > 0) {
#ifndef DBG_NO_EXACT
            pt_exact_batch(S, W, sh, wv, &stack[0][0], n_xlight, n_xtrace);
#endif
            idle_spins = 0;
            lap(prof.t_exact);
            continue;
        }
        // shaders first when a full wave of paths waits (or when it is all there is to do)
        if (ns >= P.shade_min || (ns > 0 && nt + nl == 0)) {
            const uint32_t got = pt_pop(sh.need[PT_Q_SHADE], &sh.cnt[PT_Q_SHADE], wv.nw, wv.cur[PT_Q_SHADE], true, wv.front_first);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (P.prio == 2) __builtin_amdgcn_s_setprio(2);
            int todo = 0;
            bool discarded = false;
            if (COUNT && P.trace_buf && got != PT_NONE) {
                const uint32_t tslot = pt_slot(sh, got);
                int tx, ty; bool tin; size_t toi;
                wf_slot_to_pixel(R, tslot + W.slot_base, tx, ty, tin, toi);
                if (tin && ty * R.width + tx == P.trace_pixel) {
                    const float4 *tr = wf_rec(W, tslot);
                    const uint32_t k = atomicAdd(reinterpret_cast<uint32_t *>(P.trace_buf), 1u);
                    if (4u * k + 5u <= P.trace_cap) for (int q = 0; q < 4; q++) P.trace_buf[1 + 4 * k + q] = tr[q];
                }
            }
#ifndef DBG_NO_SHADE
            if (got != PT_NONE) { PtPark pk; pk.p = (PtLdsWord)&stack[0][lane]; todo = pt_shade_lean<FEAT>(S, R, W, pt_slot(sh, got), pk, discarded); }
#endif
            n_discarded += __popcll(pt_ballot(discarded));
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            const bool next = got != PT_NONE && todo != PT_SHADE_EXACT && (todo & WF_NEXT_TRACE), with_light = next && (todo & WF_NEXT_LIGHT);
            bool wire = false; // the new ray pierces a tripwire (rt_exact.h): its closest hit is the exact role's
            if (S.n_tripwire_groups && next) {
                const float4 *nr = wf_rec(W, pt_slot(sh, got));
                const float4 n0 = nr[0], n1 = nr[1];
                wire = pt_tripwire(S, f3(n0.x, n0.y, n0.z), f3(n0.w, n1.x, n1.y));
            }
            if (next) atomicOr(&sh.pending[got >> 4], (PT_BIT_T | (with_light ? PT_BIT_L : 0u)) << ((got & 15u) * 2u));
            pt_push(sh, PT_Q_TRACE, got, next && !wire);
            pt_push(sh, PT_Q_LIGHT, got, with_light);
            pt_push(sh, PT_Q_XTRACE, got, (got != PT_NONE && todo == PT_SHADE_EXACT) || wire);
            if (got != PT_NONE && todo != PT_SHADE_EXACT) atomicAdd(&sh.cost[got >> pt_gshift(sh)], (uint32_t)PT_COST_SHADE);
            const unsigned long long done = pt_ballot(got != PT_NONE && (todo == 0 || todo == WF_PARKED)); // finished, or parked for the next phase
            if (done && lane == 0) atomicSub(&sh.cnt[PT_N_LIVE], (int)__popcll(done));
            idle_spins = 0;
            if (P.prio == 2) __builtin_amdgcn_s_setprio(0);
            if (COUNT) { prof.shade_batches++; prof.shade_items += __popcll(pt_ballot(got != PT_NONE)); }
            lap(prof.t_shade);
            continue;
        }
        if (nt + nl > 0) {
            // walkers: the kind whose backlog per walking wave (weighted by the cost of a query) is larger
            const long long wt = (long long)nt * P.cost_t * (pt_count(&sh.cnt[PT_W_LIGHT]) + 1), wl = (long long)nl * P.cost_l * (pt_count(&sh.cnt[PT_W_TRACE]) + 1);
            if (P.prio == 1) __builtin_amdgcn_s_setprio(2);
            if (nl == 0 || (nt > 0 && wt >= wl)) {
                if (lane == 0) atomicAdd(&sh.cnt[PT_W_TRACE], 1);
#ifndef DBG_NO_TRACE
                pt_trace_stint<COUNT>(S, W, sh, P, wv, stack, shade_thr, n_closest, n_nodes, n_tris, prof);
#endif
                if (lane == 0) atomicSub(&sh.cnt[PT_W_TRACE], 1);
                lap(prof.t_trace);
            } else {
                if (lane == 0) atomicAdd(&sh.cnt[PT_W_LIGHT], 1);
#ifndef DBG_NO_LIGHT
                pt_light_stint<COUNT>(S, W, sh, P, wv, stack, shade_thr, n_light, n_nodes, n_tris, prof);
#endif
                if (lane == 0) atomicSub(&sh.cnt[PT_W_LIGHT], 1);
                lap(prof.t_light);
            }
            if (P.prio == 1) __builtin_amdgcn_s_setprio(0);
            idle_spins = 0;
            if (COUNT) prof.stints++;
            continue;
        }
        if (pt_count(&sh.cnt[PT_N_LIVE]) <= 0) break;
        // paths are in flight in other waves' registers: wait for them
        __builtin_amdgcn_s_sleep(8);
        lap(prof.t_idle);
        if (++idle_spins > (1u << 24)) { gave_up = 2; break; } // safety net (seconds): never hang the GPU on a lost path; the host reports it
    }
    if (gave_up && lane == 0 && P.counters) atomicAdd(&P.counters[gave_up == 1 ? 29 : 14], 1ull);
    if (P.group_cost) { // every wave leaves the loop once the workgroup's pixels are done (or at the deadline)
        __syncthreads();
        for (uint32_t i = tid; i < n_local_groups; i += P8_THREADS) P.group_cost[sh.groups[i]] = sh.cost[i];
    }
    if (lane == 0 && P.counters) {
        if (n_closest) atomicAdd(&P.counters[0], (unsigned long long)n_closest);
        if (n_light) atomicAdd(&P.counters[1], (unsigned long long)n_light);
        if (n_discarded) atomicAdd(&P.counters[10], (unsigned long long)n_discarded);
        if (n_xtrace) atomicAdd(&P.counters[12], (unsigned long long)n_xtrace);
        if (n_xlight) atomicAdd(&P.counters[13], (unsigned long long)n_xlight);
    }
    if (COUNT && P.counters) {
        atomicAdd(&P.counters[2], n_nodes); atomicAdd(&P.counters[3], n_tris);
        atomicAdd(&P.counters[58], prof.light_hits); atomicAdd(&P.counters[59], prof.light_tests);
        if (lane == 0) { // wave-level profile, words 16..27 and 48..57
            atomicAdd(&P.counters[16], prof.t_trace); atomicAdd(&P.counters[17], prof.t_light); atomicAdd(&P.counters[18], prof.t_shade);
            atomicAdd(&P.counters[19], prof.t_exact); atomicAdd(&P.counters[20], prof.t_idle);
            atomicAdd(&P.counters[21], prof.trace_iters); atomicAdd(&P.counters[22], prof.trace_lane_iters);
            atomicAdd(&P.counters[23], prof.light_iters); atomicAdd(&P.counters[24], prof.light_lane_iters);
            atomicAdd(&P.counters[25], prof.stints); atomicAdd(&P.counters[26], prof.shade_batches); atomicAdd(&P.counters[27], prof.shade_items);
            for (int k = 0; k < 3; k++) atomicAdd(&P.counters[60 + k], prof.t_sub[0][k]);
            atomicAdd(&P.counters[63], prof.refills[0]);
            for (int w = 0; w < 2; w++) {
                for (int k = 0; k < 3; k++) atomicAdd(&P.counters[48 + 3 * w + k], prof.t_part[w][k]);
                atomicAdd(&P.counters[54 + 2 * w], prof.leaf_iters[w]); atomicAdd(&P.counters[55 + 2 * w], prof.leaf_lane_iters[w]);
            }
        }
    }
    if (P.debug && lane == 0) {
        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
        atomicMax(&P.debug[3 * blockIdx.x + 1], now);
    }
}

} // namespace dev
} // namespace rtamd
