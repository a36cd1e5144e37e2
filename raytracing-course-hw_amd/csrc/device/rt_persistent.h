// Persistent dataflow form of the hw8 replay path tracer: ONE launch renders the frame, no rounds and no global barrier.
//
// The reference's per-pixel loop (hw8/src/scene.cpp:84-177) is a chain of dependent stages per pixel — closest hit
// (bvh.h:111-142) -> shade / sample (scene.cpp:99-156) -> light-pdf sum (distributions.h:148-165) -> throughput update
// (scene.cpp:158-164) -> next bounce or next sample — and the chain of one pixel never depends on another pixel's.  The
// round-based pipeline of rt_wavefront.h runs each stage as a kernel over all pixels and pays a machine-wide drain at every
// kernel boundary (the launch ends with its longest walk).  Here the dependency is kept PER PATH:
//
//   * one 1024-thread workgroup per CU owns a fixed, interleaved share of the pixel slots (8x8 sub-tiles dealt round-robin
//     to the workgroups); the path records stay in HBM in the layout of rt_wavefront.h, but only this workgroup touches
//     them, so hand-offs between its waves need workgroup-scope ordering only (one CU, one L1: `s_waitcnt` + LDS);
//   * the stage a path waits for is one bit per path in an LDS bitmap (need_trace / need_light / need_shade / ...); a wave
//     that wants work claims set bits with LDS atomics (one word per lane, rotating cursor: round-robin service, no
//     capacity limit, no ring to overflow);
//   * every wave picks a role when it is idle: closest-hit walker, light-sum walker or shader, by the populations of the
//     bitmaps; walkers keep their lanes full by refilling idle lanes from the bitmap; a shader takes up to 64 paths,
//     finishes their pending bounce, shades the new hit and sets the bits of what each path needs next;
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
// ref_light_pdf_sum).  Pixels then match the reference's also where a ray grazes a box corner.
#pragma once
#include "rt_wavefront.h"

namespace rtamd {
namespace dev {

#define PT_THREADS 1024
#define PT_WAVES 16
// Experiment: -DPT_TREELET=511 keeps the top 511 nodes of the scene tree (breadth-first, 32,704 B) in LDS; the
// bitmaps then hold 8,192 paths per workgroup instead of 32,768 so that stacks + bitmaps + treelet fill the 160 KB exactly.
#ifndef PT_TREELET
#define PT_TREELET 0
#endif
#if PT_TREELET
#define PT_MAX_PATHS 8192
#else
#define PT_MAX_PATHS 32768            // paths per workgroup (bitmap capacity in LDS)
#endif
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

struct PtShared {
    uint32_t stack[PT_WAVES][WF_STACK][64];   // per-lane traversal stack columns, one area per wave
    uint32_t need[5][PT_NW];
    uint32_t pending[PT_NW * 2];              // 2 bits per path
    uint32_t groups[PT_MAX_PATHS / 64];       // local 64-slot group -> group of the pass (8x8 sub-tile)
    uint32_t cost[PT_MAX_PATHS / 64];         // shaded hits per local group in this launch: the load measure the frame is re-dealt by
    int cnt[16];
#if PT_TREELET
    float4 treelet[PT_TREELET][4];
#endif
};

struct PtParams {
    uint32_t n_groups;                // 64-slot groups (8x8 sub-tiles) of this pass
    // Which groups a workgroup owns: group_ids[group_ofs[b] .. group_ofs[b + 1]) when group_ofs is set (the host's re-deal after
    // the first phase of a frame), else b, b + n_blocks, b + 2 n_blocks, ...; `resume` = the paths carry on from their records
    // (a later phase) instead of being seeded; group_cost[g] receives the number of hits shaded for group g in this launch.
    const uint32_t *group_ofs, *group_ids;
    uint32_t *group_cost;
    uint32_t resume;
    int refill, leaf_batch;           // as in rt_wavefront.h (leaf_batch = batch | share << 16)
    int shade_thr0, shade_thr_step;   // wave w stops refilling its walkers when need_shade holds >= thr0 + w * step paths
    int cost_t, cost_l;               // relative cost of a closest-hit / light query (walker split)
    int prio;                         // experiment: 1 = walker stints run at raised wave priority (s_setprio 2), 2 = shader batches do
    unsigned long long deadline_ticks; // 100 MHz ticks a wave may spend in this launch before it gives up (error)
    unsigned long long *counters;     // [0] closest-hit queries, [1] light queries, [2] node visits, [3] triangle tests, [10] discarded speculative hits, [12] exact closest hits, [13] exact light sums, [14] waves that gave up waiting (error)
    unsigned long long *debug;        // nullable: per workgroup {start time, exit time of its last wave (100 MHz ticks), paths}
    // COUNT builds, RTAMD_TRACE_PIXEL: every hit record the shader consumes for pixel trace_pixel (= y * width + x) is appended as
    // four float4 (r0..r3 of the path record: ray, hit, packed word); word 0 of trace_buf counts the entries
    float4 *trace_buf; uint32_t trace_cap; int32_t trace_pixel;
};

// COUNT builds only: where a wave's time goes (shader-clock cycles per role) and how full its walker iterations are
struct PtProf {
    unsigned long long t_trace = 0, t_light = 0, t_shade = 0, t_exact = 0, t_idle = 0;
    unsigned long long trace_iters = 0, trace_lane_iters = 0, light_iters = 0, light_lane_iters = 0, stints = 0, shade_batches = 0, shade_items = 0;
};

// wave-uniform state
struct PtWave {
    uint32_t nw, n_local, n_blocks, block;
    uint32_t cur[5];
};

template <class SH> RT_DEV uint32_t pt_slot(const SH &sh, uint32_t l) { return (sh.groups[l >> 6] << 6) | (l & 63u); }
RT_DEV int pt_count(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } // a fresh LDS read each time

// Hands paths to the lanes that want one.  The wave reads 64 bitmap words at once (lane i: word cursor + i), then takes whole
// words in order — ONE atomicAnd per word by lane 0 — and deals the claimed bits to the wanting lanes by rank, so a wave that
// wants 64 paths from a dense queue gets the two words of one 8x8 sub-tile (coherent rays) for two LDS atomics.  A word that
// holds more paths than are wanted keeps its upper bits and the cursor stays on it, so the next request starts there (no
// path is passed over).  Returns the local path index or PT_NONE.
RT_DEV uint32_t pt_pop(uint32_t *bm, int *cnt, const uint32_t nw, uint32_t &cursor, bool want) {
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long wantmask = __ballot(want);
    const int need = __popcll(wantmask);
    const int my_rank = __popcll(wantmask & ((1ull << lane) - 1ull));
    uint32_t got = PT_NONE;
    int have = 0;
    for (uint32_t swept = 0; swept < nw && have < need; swept += 64u) {
        uint32_t w = cursor + lane;
        bool valid = true;
        if (nw >= 64u) { if (w >= nw) w -= nw; }
        else { valid = lane < nw; w = w % nw; }
        const uint32_t v = valid ? bm[w] : 0u;
        unsigned long long nz = __ballot(v != 0u);
        uint32_t next_cursor = cursor + 64u;
        while (nz && have < need) {
            const int j = __ffsll((long long)nz) - 1;
            nz &= nz - 1ull;
            const uint32_t vj = (uint32_t)__shfl((int)v, j), wj = (uint32_t)__shfl((int)w, j);
            uint32_t rest = vj;                                   // the lowest (need - have) set bits of vj
            for (int n = need - have; n > 0 && rest; n--) rest &= rest - 1u;
            const uint32_t take = vj & ~rest;
            uint32_t old = 0;
            if (lane == 0) old = atomicAnd(&bm[wj], ~take);
            old = (uint32_t)__shfl((int)old, 0) & take;           // the bits this wave really claimed
            const int c = __popc(old);
            if (want && my_rank >= have && my_rank < have + c) {
                uint32_t bits = old;
                for (int k = my_rank - have; k > 0; k--) bits &= bits - 1u;
                got = wj * 32u + (uint32_t)__ffs((int)bits) - 1u;
            }
            have += c;
            if (rest) next_cursor = wj;                          // paths left in this word: come back to it first
            else if (have >= need) next_cursor = wj + 1u;
        }
        cursor = next_cursor;
        if (cursor >= nw) cursor %= nw;
    }
    if (have && lane == 0) atomicSub(cnt, have);
    return got;
}

// Sets the bit of path l in queue q for the lanes with `doit` (wave-uniform call).
template <class SH> RT_DEV void pt_push(SH &sh, int q, uint32_t l, bool doit) {
    if (doit) atomicOr(&sh.need[q][l >> 5], 1u << (l & 31u));
    const unsigned long long m = __ballot(doit);
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
    bool active = false, refill_ok = true;
    uint32_t l = 0, slot = 0, cur = 0, hit = WF_MISS, fin = PT_NONE;
    int sp = 0;
    F3 o = f3(0.f, 0.f, 0.f), d = f3(0.f, 0.f, 1.f);
    RayInv ray = make_ray_inv(o, d);
    float best_t = RT_T_MAX, best_u = 0.f, best_v = 0.f;
    // Boxes are pruned, and farther hits dropped, only beyond cull_t = best_t + the look-behind of rt_exact.h: the runner-up of the
    // best hit must be SEEN, whatever tree the walk uses, to decide at the end of the walk whether the exact walk is needed.
    float cull_t = RT_T_MAX, t2 = 2.f * RT_T_MAX, h_ray = 0.f; // h_ray: absolute part of the look-behind (pt_look_behind)
    auto store_hit = [&]() { // the gate (pt_shade_item) decides with the runner-up's t whether this hit needs the exact walk
        wf_rec(W, slot)[2] = make_float4(best_t, best_u, best_v, __uint_as_float(S.exact_boxes && hit != WF_MISS ? hit | pt_gap_code(best_t, t2) : hit));
    };
    for (;;) {
        const unsigned long long idle = __ballot(!active);
        if (idle && (__popcll(idle) >= P.refill || idle == ~0ull)) {
            // Hand-off point.  Finished lanes are published here and not the moment they finish: the release (a wait for the
            // wave's outstanding record stores) is paid once per refill, when the stores have long landed, not once per walk.
            if (__ballot(fin != PT_NONE)) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                pt_complete(sh, fin, PT_BIT_T, fin != PT_NONE);
                fin = PT_NONE;
            }
            if (!refill_ok) {}
            else if (pt_count(&sh.cnt[PT_Q_SHADE]) >= shade_thr) refill_ok = false;      // shaders are behind: drain, then help them
            else if (pt_count(&sh.cnt[PT_Q_TRACE]) > 0) {
                const uint32_t got = pt_pop(sh.need[PT_Q_TRACE], &sh.cnt[PT_Q_TRACE], wv.nw, wv.cur[PT_Q_TRACE], !active);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                n_queries += __popcll(__ballot(got != PT_NONE));
                if (got != PT_NONE) {
                    l = got; slot = pt_slot(sh, l);
                    const float4 *r = wf_rec(W, slot);
                    float4 q0 = r[0], q1 = r[1];
                    o = f3(q0.x, q0.y, q0.z); d = f3(q0.w, q1.x, q1.y);
                    ray = make_ray_inv(o, d);
                    h_ray = S.exact_boxes ? pt_look_behind_abs(d, S.box_c2) : 0.f;
                    cur = 0; sp = 0; hit = WF_MISS; best_t = RT_T_MAX; cull_t = RT_T_MAX; t2 = 2.f * RT_T_MAX; best_u = 0.f; best_v = 0.f;
                    active = true;
                }
            }
        }
        const unsigned long long m_active = __ballot(active);
        if (!m_active) break;
        const int lb = min(P.leaf_batch & 255, (__popcll(m_active) * (P.leaf_batch >> 16) + 255) >> 8);
        for (;;) { // phase 1: inner nodes
            const bool inner = active && !(cur & RT_LEAF_BIT);
            if (!__ballot(inner) || __popcll(__ballot(active && (cur & RT_LEAF_BIT))) >= lb) break;
            if (COUNT) { prof.trace_iters++; prof.trace_lane_iters += __popcll(__ballot(inner)); }
            if (inner) {
#if PT_TREELET
                const float4 *q = cur < (uint32_t)PT_TREELET ? (const float4 *)sh.treelet[cur] : reinterpret_cast<const float4 *>(S.nodes + cur);
#else
                const float4 *q = reinterpret_cast<const float4 *>(S.nodes + cur);
#endif
                float4 lo0 = q[0], hi0 = q[1], lo1 = q[2], hi1 = q[3];
                if (COUNT) n_nodes++;
                float n0, n1;
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
            }
        }
        if (active && (cur & RT_LEAF_BIT)) { // phase 2: leaves
            if (cur != RT_EMPTY_LEAF) {
                uint32_t i = cur & ~RT_LEAF_BIT;
                for (;;) {
                    TriIsect T = load_isect(S.tri_walk + i);
                    if (COUNT) n_tris++;
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
                    if (T.pad & 1u) break;
                    i++;
                }
            }
            if (sp == 0) {
                store_hit();
                active = false; fin = l;
            } else cur = stack[--sp][lane];
        }
    }
}

// ---- light-sum walker (wf_light_loop_lean of rt_wavefront.h fed from the need_light bitmap) ----------------------------------
template <bool COUNT>
RT_DEV void pt_light_stint(const SceneView &S, const WfView &W, PtShared &sh, const PtParams &P, PtWave &wv, uint32_t (*stack)[64],
                           const int shade_thr, uint32_t &n_queries, unsigned long long &n_nodes, unsigned long long &n_tris, PtProf &prof) {
    const int lane = threadIdx.x & 63;
    bool active = false, overflow = false, refill_ok = true;
    uint32_t l = 0, slot = 0, cur = 0, fin = PT_NONE, slow = PT_NONE;
    int sp = 0, k = 0;
    F3 o = f3(0.f, 0.f, 0.f), d = f3(0.f, 0.f, 1.f);
    RayInv ray = make_ray_inv(o, d);
    auto finish = [&]() {
        active = false;
        if (overflow) { slow = l; return; }
        float v = 0.f;
        if (k == 1) v = __uint_as_float(stack[WF_STACK - 2][lane]);
        else if (k == 2) v = __uint_as_float(stack[WF_STACK - 2][lane]) + __uint_as_float(stack[WF_STACK - 4][lane]);
        else if (k > 2) { // the reference's association of the additions, see wf_light_loop_lean
            const uint32_t nl = S.n_lights;
            for (int j = 1; j < k; j++) {
                uint32_t a0 = stack[WF_STACK - 1 - 2 * (j - 1)][lane], b0 = stack[WF_STACK - 1 - 2 * j][lane];
                uint32_t len = b0 - a0;
                uint32_t lv = 31u - (uint32_t)__clz((int)len);
                uint16_t m0 = S.light_sep[(size_t)lv * nl + a0], m1 = S.light_sep[(size_t)lv * nl + (b0 - (1u << lv))];
                stack[j - 1][lane] = m0 < m1 ? m0 : m1;
            }
            for (int n = k; n > 1; n--) {
                int best = 1;
                uint32_t bd = stack[0][lane];
                for (int i = 2; i < n; i++) { uint32_t di = stack[i - 1][lane]; if (di > bd) { bd = di; best = i; } }
                float merged = __uint_as_float(stack[WF_STACK - 2 - 2 * (best - 1)][lane]) + __uint_as_float(stack[WF_STACK - 2 - 2 * best][lane]);
                stack[WF_STACK - 2 - 2 * (best - 1)][lane] = __float_as_uint(merged);
                for (int i = best; i < n - 1; i++) {
                    stack[WF_STACK - 2 - 2 * i][lane] = stack[WF_STACK - 2 - 2 * (i + 1)][lane];
                    stack[i - 1][lane] = stack[i][lane];
                }
            }
            v = __uint_as_float(stack[WF_STACK - 2][lane]);
        }
        int depth = (int)(__float_as_uint(reinterpret_cast<const float *>(wf_rec(W, slot) + 3)[3]) & 15u);
        float *pdf = reinterpret_cast<float *>(wf_entry(W, slot, depth)) + 3;
        *pdf = *pdf + v / (float)S.n_lights;                                  // distributions.h:123,273
        fin = l;
    };
    for (;;) {
        const unsigned long long idle = __ballot(!active);
        if (idle && (__popcll(idle) >= P.refill || idle == ~0ull)) {
            if (__ballot(fin != PT_NONE || slow != PT_NONE)) { // hand-off point, see pt_trace_stint
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                pt_complete(sh, fin, PT_BIT_L, fin != PT_NONE);
                pt_push(sh, PT_Q_XLIGHT, slow, slow != PT_NONE);
                fin = PT_NONE; slow = PT_NONE;
            }
            if (!refill_ok) {}
            else if (pt_count(&sh.cnt[PT_Q_SHADE]) >= shade_thr) refill_ok = false;
            else if (pt_count(&sh.cnt[PT_Q_LIGHT]) > 0) {
                const uint32_t got = pt_pop(sh.need[PT_Q_LIGHT], &sh.cnt[PT_Q_LIGHT], wv.nw, wv.cur[PT_Q_LIGHT], !active);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                n_queries += __popcll(__ballot(got != PT_NONE));
                if (got != PT_NONE) {
                    l = got; slot = pt_slot(sh, l);
                    const float4 *r = wf_rec(W, slot);
                    float4 q0 = r[0], q1 = r[1];
                    o = f3(q0.x, q0.y, q0.z); d = f3(q0.w, q1.x, q1.y);
                    ray = make_ray_inv(o, d);
                    cur = 0; sp = 0; k = 0; overflow = false;
                    active = true;
                }
            }
        }
        const unsigned long long m_active = __ballot(active);
        if (!m_active) break;
        const int lb = min(P.leaf_batch & 255, (__popcll(m_active) * (P.leaf_batch >> 16) + 255) >> 8);
        for (;;) { // phase 1: inner nodes
            const bool inner = active && !(cur & RT_LEAF_BIT);
            if (!__ballot(inner) || __popcll(__ballot(active && (cur & RT_LEAF_BIT))) >= lb) break;
            if (COUNT) { prof.light_iters++; prof.light_lane_iters += __popcll(__ballot(inner)); }
            if (inner) {
                const float4 *q = reinterpret_cast<const float4 *>(S.light_nodes + cur);
                float4 lo0 = q[0], hi0 = q[1], lo1 = q[2], hi1 = q[3];
                if (COUNT) n_nodes++;
                float n0, n1;
                bool h0 = slab_test(lo0, hi0, ray, RT_T_MAX, n0);
                bool h1 = slab_test(lo1, hi1, ray, RT_T_MAX, n1);
                uint32_t c0 = __float_as_uint(lo0.w), c1 = __float_as_uint(lo1.w);
                if (h0 & h1) { stack[sp++][lane] = c1; cur = c0; if (sp + 2 * k >= WF_STACK) overflow = true; }
                else if (h0) cur = c0;
                else if (h1) cur = c1;
                else if (sp == 0) finish();
                else cur = stack[--sp][lane];
            }
        }
        if (active && (cur & RT_LEAF_BIT)) { // phase 2: leaves
            if (cur != RT_EMPTY_LEAF) {
                uint32_t i = cur & ~RT_LEAF_BIT;
                for (;;) {
                    bool last, robust;
                    if (COUNT) n_tris++;
                    float term = pt_light_pdf_one(S, S.lights + i, o, d, last, robust);
                    if (term != 0.f) { // a hit (a miss contributes exactly 0, and adding 0 changes nothing)
                        if (!robust || k >= WF_MAX_LIGHT_HITS || sp + 2 * k + 2 >= WF_STACK) overflow = true;
                        else { stack[WF_STACK - 1 - 2 * k][lane] = i; stack[WF_STACK - 2 - 2 * k][lane] = __float_as_uint(term); k++; }
                    }
                    if (last) break;
                    i++;
                }
            }
            if (sp == 0) finish();
            else cur = stack[--sp][lane];
        }
    }
}

// ---- the kernel -----------------------------------------------------------------------------------------------------------
template <bool COUNT, int FEAT>
__global__ __launch_bounds__(PT_THREADS) void pt_persistent_kernel(SceneView S, RenderView R, WfView W, PtParams P) {
    __shared__ PtShared sh;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    PtWave wv;
    wv.n_blocks = gridDim.x; wv.block = blockIdx.x;
    const uint32_t first_group = P.group_ofs ? P.group_ofs[wv.block] : 0u;
    const uint32_t n_local_groups = P.group_ofs ? P.group_ofs[wv.block + 1u] - first_group
                                                : (P.n_groups > wv.block ? (P.n_groups - wv.block + wv.n_blocks - 1u) / wv.n_blocks : 0u);
    wv.n_local = n_local_groups * 64u;
    wv.nw = n_local_groups * 2u;
    if (wv.n_local == 0u) return;
    if (P.debug && tid == 0) { P.debug[3 * blockIdx.x] = __builtin_amdgcn_s_memrealtime(); P.debug[3 * blockIdx.x + 2] = wv.n_local; }
    for (int q = 0; q < 5; q++) wv.cur[q] = (wave * 64u) % wv.nw;

    // ---- init: seed every pixel of this workgroup, first camera ray (wf_init_kernel of rt_wavefront.h) -----------------------
    for (uint32_t i = tid; i < wv.nw; i += PT_THREADS) { sh.need[0][i] = 0; sh.need[1][i] = 0; sh.need[2][i] = 0; sh.need[3][i] = 0; sh.need[4][i] = 0; }
    for (uint32_t i = tid; i < 2u * wv.nw; i += PT_THREADS) sh.pending[i] = 0;
    for (uint32_t i = tid; i < n_local_groups; i += PT_THREADS) { sh.groups[i] = P.group_ofs ? P.group_ids[first_group + i] : i * wv.n_blocks + wv.block; sh.cost[i] = 0; }
    if (tid < 16u) sh.cnt[tid] = 0;
#if PT_TREELET
    for (uint32_t i = tid; i < (uint32_t)PT_TREELET * 4u; i += PT_THREADS)
        sh.treelet[i >> 2][i & 3u] = (i >> 2) < S.n_nodes ? reinterpret_cast<const float4 *>(S.nodes + (i >> 2))[i & 3u] : make_float4(0.f, 0.f, 0.f, 0.f);
#endif
    __syncthreads();
    for (uint32_t base = 0; base < wv.n_local; base += PT_THREADS) {
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
        const unsigned long long m = __ballot(started);
        if (m && lane == 0) atomicAdd(&sh.cnt[PT_N_LIVE], (int)__popcll(m));
        pt_push(sh, PT_Q_TRACE, l, started);
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
    PtProf prof;
    unsigned long long t_mark = COUNT ? __builtin_amdgcn_s_memtime() : 0ull;
    auto lap = [&](unsigned long long &acc) { if (COUNT) { const unsigned long long t = __builtin_amdgcn_s_memtime(); acc += t - t_mark; t_mark = t; } };
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        if (__builtin_amdgcn_s_memrealtime() - t_start > P.deadline_ticks) { // safety net: never hang the GPU; the host reports the error
            if (lane == 0 && P.counters) atomicAdd(&P.counters[14], 1ull);
            break;
        }
        const int ns = pt_count(&sh.cnt[PT_Q_SHADE]), nt = pt_count(&sh.cnt[PT_Q_TRACE]), nl = pt_count(&sh.cnt[PT_Q_LIGHT]);
        const int nx = pt_count(&sh.cnt[PT_Q_XLIGHT]) + pt_count(&sh.cnt[PT_Q_XTRACE]);
        if (nx > 0) {
            // exact role: one lane per query, the reference's own box arithmetic over the reference trees
            uint32_t xstack[RT_STACK_SIZE];
            uint32_t got = pt_pop(sh.need[PT_Q_XLIGHT], &sh.cnt[PT_Q_XLIGHT], wv.nw, wv.cur[PT_Q_XLIGHT], true);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (got != PT_NONE) {
                const uint32_t slot = pt_slot(sh, got);
                const float4 *r = wf_rec(W, slot);
                float4 q0 = r[0], q1 = r[1];
                const F3 x = f3(q0.x, q0.y, q0.z), d = f3(q0.w, q1.x, q1.y);
                float v;
                if (S.exact_boxes) v = ref_light_pdf_sum(S, x, d, xstack);
                else { Counters c; c.closest = c.lightq = c.nodes = c.tris = 0; v = light_pdf_sum<false>(S, x, d, xstack, c); }
                int depth = (int)(__float_as_uint(r[3].w) & 15u);
                float *pdf = reinterpret_cast<float *>(wf_entry(W, slot, depth)) + 3;
                *pdf = *pdf + v / (float)S.n_lights;
            }
            n_xlight += __popcll(__ballot(got != PT_NONE));
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            pt_complete(sh, got, PT_BIT_L, got != PT_NONE);
            got = pt_pop(sh.need[PT_Q_XTRACE], &sh.cnt[PT_Q_XTRACE], wv.nw, wv.cur[PT_Q_XTRACE], true);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (got != PT_NONE) {
                const uint32_t slot = pt_slot(sh, got);
                float4 *r = wf_rec(W, slot);
                float4 q0 = r[0], q1 = r[1];
                float bt, bu, bv; uint32_t hit;
                ref_closest_hit(S, f3(q0.x, q0.y, q0.z), f3(q0.w, q1.x, q1.y), xstack, bt, bu, bv, hit);
                r[2] = make_float4(bt, bu, bv, __uint_as_float(hit));
                float *pk = reinterpret_cast<float *>(r + 3) + 3;
                *pk = __uint_as_float(__float_as_uint(*pk) | WF_VERIFIED_BIT);
            }
            n_xtrace += __popcll(__ballot(got != PT_NONE));
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            pt_push(sh, PT_Q_SHADE, got, got != PT_NONE);
            idle_spins = 0;
            lap(prof.t_exact);
            continue;
        }
        // shaders first when a full wave of paths waits (or when it is all there is to do)
        if (ns >= 64 || (ns > 0 && nt + nl == 0)) {
            const uint32_t got = pt_pop(sh.need[PT_Q_SHADE], &sh.cnt[PT_Q_SHADE], wv.nw, wv.cur[PT_Q_SHADE], true);
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
            if (got != PT_NONE) todo = pt_shade_item<FEAT>(S, R, W, pt_slot(sh, got), discarded);
            n_discarded += __popcll(__ballot(discarded));
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            const bool next = got != PT_NONE && (todo & WF_NEXT_TRACE), with_light = next && (todo & WF_NEXT_LIGHT);
            if (next) atomicOr(&sh.pending[got >> 4], (PT_BIT_T | (with_light ? PT_BIT_L : 0u)) << ((got & 15u) * 2u));
            pt_push(sh, PT_Q_TRACE, got, next);
            pt_push(sh, PT_Q_LIGHT, got, with_light);
            pt_push(sh, PT_Q_XTRACE, got, got != PT_NONE && todo == PT_SHADE_EXACT);
            if (got != PT_NONE && todo != PT_SHADE_EXACT) atomicAdd(&sh.cost[got >> 6], 1u);
            const unsigned long long done = __ballot(got != PT_NONE && (todo == 0 || todo == WF_PARKED)); // finished, or parked for the next phase
            if (done && lane == 0) atomicSub(&sh.cnt[PT_N_LIVE], (int)__popcll(done));
            idle_spins = 0;
            if (P.prio == 2) __builtin_amdgcn_s_setprio(0);
            if (COUNT) { prof.shade_batches++; prof.shade_items += __popcll(__ballot(got != PT_NONE)); }
            lap(prof.t_shade);
            continue;
        }
        if (nt + nl > 0) {
            // walkers: the kind whose backlog per walking wave (weighted by the cost of a query) is larger
            const long long wt = (long long)nt * P.cost_t * (pt_count(&sh.cnt[PT_W_LIGHT]) + 1), wl = (long long)nl * P.cost_l * (pt_count(&sh.cnt[PT_W_TRACE]) + 1);
            if (P.prio == 1) __builtin_amdgcn_s_setprio(2);
            if (nl == 0 || (nt > 0 && wt >= wl)) {
                if (lane == 0) atomicAdd(&sh.cnt[PT_W_TRACE], 1);
                pt_trace_stint<COUNT>(S, W, sh, P, wv, stack, shade_thr, n_closest, n_nodes, n_tris, prof);
                if (lane == 0) atomicSub(&sh.cnt[PT_W_TRACE], 1);
                lap(prof.t_trace);
            } else {
                if (lane == 0) atomicAdd(&sh.cnt[PT_W_LIGHT], 1);
                pt_light_stint<COUNT>(S, W, sh, P, wv, stack, shade_thr, n_light, n_nodes, n_tris, prof);
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
        if (++idle_spins > (1u << 24)) { // safety net (seconds): never hang the GPU on a lost path; the host reports it
            if (lane == 0 && P.counters) atomicAdd(&P.counters[14], 1ull);
            break;
        }
    }
    if (P.group_cost) { // every wave leaves the loop once the workgroup's pixels are done (or at the deadline)
        __syncthreads();
        for (uint32_t i = tid; i < n_local_groups; i += PT_THREADS) P.group_cost[sh.groups[i]] = sh.cost[i];
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
        if (lane == 0) { // wave-level profile, words 16..27
            atomicAdd(&P.counters[16], prof.t_trace); atomicAdd(&P.counters[17], prof.t_light); atomicAdd(&P.counters[18], prof.t_shade);
            atomicAdd(&P.counters[19], prof.t_exact); atomicAdd(&P.counters[20], prof.t_idle);
            atomicAdd(&P.counters[21], prof.trace_iters); atomicAdd(&P.counters[22], prof.trace_lane_iters);
            atomicAdd(&P.counters[23], prof.light_iters); atomicAdd(&P.counters[24], prof.light_lane_iters);
            atomicAdd(&P.counters[25], prof.stints); atomicAdd(&P.counters[26], prof.shade_batches); atomicAdd(&P.counters[27], prof.shade_items);
        }
    }
    if (P.debug && lane == 0) {
        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
        atomicMax(&P.debug[3 * blockIdx.x + 1], now);
    }
}

} // namespace dev
} // namespace rtamd
