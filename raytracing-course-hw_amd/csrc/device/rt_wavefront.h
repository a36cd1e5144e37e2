// Wavefront (multi-kernel) form of the hw8 replay path tracer.
//
// The reference's per-pixel loop (hw8/src/scene.cpp:84-177) is a chain of dependent stages per pixel
// (closest hit -> shade/sample -> light-pdf traversal -> throughput update -> next bounce or next
// sample).  A pixel's samples must stay sequential (they share one minstd_rand stream), but pixels
// are independent, so every pixel carries its path state in HBM and the stages run as kernels over
// queues of the pixels that are at that stage.  Two launches per round:
//
//   traverse(r)  one persistent launch, two wave populations sharing the CUs:
//                  * closest hit for every ray of q_trace(r)           (bvh.h:111-142)
//                  * all-hits light sum for every ray of q_light(r)    (FiguresMix::getTotalPdf)
//                The sampled direction of a bounce is both its light-pdf query and the next bounce's ray, and
//                tracing draws no random numbers, so the next closest hit is traced SPECULATIVELY next to the pdf
//                sum of the bounce that produced it; the clamp test (scene.cpp:161-163) that needs the pdf may
//                then discard it.  The split between the populations follows the queue lengths; a wave whose
//                queue has run dry takes chunks from the other queue's dynamic tail.
//   shade(r)     per pixel of q_trace(r): finish the pending bounce (Mix::pdf, throughput, clamp hack), then
//                attribute/texture fetch, Mix::sample, BRDF for the new hit; ends paths that miss / clamp /
//                early-out (backward fold, next camera ray or pixel write) and queues the next ray.
//
// A pixel advances one bounce per round; spp * ray_depth rounds finish every pixel.  The traversal loops are
// lean (few VGPRs, LDS stacks, all 64 lanes doing the same kind of work) and keep their lanes busy by pulling
// the next ray as soon as one finishes.  Arithmetic is the same bit-exact code as the single-kernel path.
//
// HBM layout per pixel slot: one 64-byte record (ray, hit, RNG, accumulator) followed by ray_depth 32-byte
// stack entries, contiguous (256 B per slot at depth 6), so a scattered access still moves whole cache
// lines and the backward fold at the end of a path reads one contiguous run.
#pragma once
#include "rt_exact.h"

namespace rtamd {
namespace dev {

// R0 record (4 x float4 per slot):
//   q0 = o.xyz, d.x        q1 = d.y, d.z, rng_x, rng_saved
//   q2 = t, u, v, hit      q3 = accum.xyz, packed{depth:4, has_saved:1, pending:1, sample:26}
// hit: 0xFFFFFFFF = miss, else triangle index | inside << 30.  pending = the bounce at `depth` still waits for its
// Mix::pdf / clamp step (its stack entry is filled, the ray in q0/q1 is the one it sampled).
// Stack entry (2 x float4): E0 = emission.xyz, pdf (cosine + vndf terms; the light loop adds its term)
//                           E1 = brdf (then mult).xyz, dot(d, n_s)
struct WfView {
    float4 *r0;             // per slot: `stride` float4 = WF_REC_BASE (R0 record) + 2 per stack level, contiguous
    uint32_t stride;
    uint32_t *q_trace[2];
    uint32_t *q_light;
    uint32_t *ctr;          // per round r, WF_CTR words: +0 trace count, +1 light count, +2 trace head, +3 light head, +4 slow-light count
    uint32_t *q_slow;       // light queries of this round that the lean loop hands to wf_light_exact_kernel
    uint32_t n_slots;
    uint32_t *ovf;          // SPILL variant only: WF_OVF stack entries per persistent thread beyond the WF_STACK entries in LDS
    uint32_t slot_base;     // this pipeline's first path slot (the frame's slots are cut into independent pipelines, one per stream)
};

RT_DEV float4 *wf_rec(const WfView &W, uint32_t slot) { return W.r0 + (size_t)slot * W.stride; }
#define WF_REC_BASE 4          // float4 in front of the per-level entries: q0..q3 (16 float4 = two 128-byte lines per path at depth 6)
RT_DEV float4 *wf_entry(const WfView &W, uint32_t slot, int level) { return W.r0 + ((size_t)slot * W.stride + (uint32_t)(WF_REC_BASE + 2 * level)); } // one 64-bit multiply-add of 32-bit operands

#define WF_CTR 8               // counter words per round
#ifndef WF_STACK
#define WF_STACK 30            // LDS traversal stack entries per lane: 30 KB per block, so that five blocks are resident per CU
#endif
#define WF_OVF 96              // deeper entries of the SPILL kernel variant (trees deeper than WF_STACK) live in global memory

#define WF_PENDING_BIT 32u
RT_DEV uint32_t wf_pack(int depth, bool has_saved, uint32_t sample, bool pending = false) {
    return (uint32_t)depth | (has_saved ? 16u : 0u) | (pending ? WF_PENDING_BIT : 0u) | (sample << 6);
}

// `gslot` = W.slot_base + the pipeline-local slot that indexes the path state
RT_DEV void wf_slot_to_pixel(const RenderView &R, uint32_t gslot, int &x, int &y, bool &inside, size_t &out_index) {
    slot_to_pixel(R, R.streams > 1 ? gslot % R.n_pixslots : gslot, x, y, inside, out_index);
}

// ---- queue append, aggregated per workgroup -------------------------------------------------------------
// One global atomic on a single address retires at ~88 per microsecond (MI355X_MICROARCH.md "dequeue"), which
// would bound a kernel that appends once per wave.  Appends therefore go to an LDS staging buffer (LDS atomic
// per wave) and reach the global queue with ONE atomic per ~2k items, written coalesced.
#define WF_BUF 2048
struct Pusher { uint32_t *buf; uint32_t *cnt; };
RT_DEV void wf_push(const Pusher &p, uint32_t slot) {
    unsigned long long mask = __ballot(1);
    int lane = threadIdx.x & 63;
    int leader = __ffsll((long long)mask) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(p.cnt, (uint32_t)__popcll(mask));
    base = __shfl(base, leader);
    uint32_t rank = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    p.buf[base + rank] = slot;
}
// Called by every thread of the block, after a __syncthreads() that follows the last push.
RT_DEV void wf_flush(const Pusher &p, uint32_t *gqueue, uint32_t *gcount, uint32_t *shared_base) {
    uint32_t n = *p.cnt;
    if (n) {
        if (threadIdx.x == 0) *shared_base = atomicAdd(gcount, n);
        __syncthreads();
        uint32_t gb = *shared_base;
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) gqueue[gb + i] = p.buf[i];
        __syncthreads();
        if (threadIdx.x == 0) *p.cnt = 0;
    }
    __syncthreads();
}

// RT_FLAG_SAMPLE_SEEDS: the engine of camera sample `sample_of_stream` of the slot's stream.  Consecutive minstd seeds give
// correlated first draws, so the (pixel, sample) pair goes through a 32-bit mixer (the finalizer of MurmurHash3) first.
RT_DEV void wf_sample_seed(const RenderView &R, Rng &rng, uint32_t gslot, int x, int y, uint32_t sample_of_stream) {
    const uint32_t stream = R.streams > 1 ? gslot / R.n_pixslots : 0u;
    const uint32_t global_sample = sample_of_stream * (uint32_t)(R.streams > 1 ? R.streams : 1) + stream;
    uint32_t h = ((uint32_t)(y * R.width + x)) * R.total_samples + global_sample;
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    rng_seed(rng, h);
}

RT_DEV void wf_camera_ray(const SceneView &S, const RenderView &R, Rng &rng, int x, int y, F3 &o, F3 &d) {
    float nx = (float)x + rng_u01(rng);                             // scene.cpp:172-173
    float ny = (float)y + rng_u01(rng);
    float cx = R.tan_fov_x * (2 * nx / (float)R.width - 1);         // scene.cpp:183-184
    float cy = S.tan_fov_y * (2 * ny / (float)R.height - 1);
    o = f3(S.cam_pos);
    d = normalize(cx * f3(S.cam_right) - cy * f3(S.cam_up) + f3(S.cam_fwd));
}

// Russian roulette (throughput mode, RT_FLAG_RUSSIAN_ROULETTE): survival probability of the bounce at `depth` whose throughput factor is `mult`.
// q = the largest component of the path's ACCUMULATED throughput up to and including this bounce (the factors of the earlier levels are in
// their entries; survivors carry 1 / q, so the accumulated throughput of a path that played before hovers around 1), between 0.25 and 1.  Per-bounce
// throughput alone (round 2: q = max component of `mult`, floor 0.05) kills bright paths as readily as dim ones: −28 % queries for 2.7x the variance.
RT_DEV float wf_roulette_q(const WfView &W, uint32_t slot, int depth, F3 mult) {
    F3 beta = mult;
    for (int b = 0; b < depth; b++) {
        const float4 e1 = wf_entry(W, slot, b)[1];
        beta = beta * f3(e1.x, e1.y, e1.z);
    }
    return fminf(1.f, fmaxf(0.25f, fmaxf(beta.x, fmaxf(beta.y, beta.z))));
}

// End of one camera sample: fold e + m*(inner) backwards (scene.cpp:164), add to the pixel sum
// (scene.cpp:174), then either start the next sample (returns WF_NEXT_TRACE: the slot holds a new camera ray that wants
// tracing; WF_PARKED instead when that sample belongs to the next phase of the frame, R.sample_stop) or write the finished
// pixel (returns 0).  The pixel sum is read here, not carried through the shading code.
#define WF_NEXT_TRACE 1
#define WF_NEXT_LIGHT 2
#define WF_PARKED 8
RT_DEV int wf_finish_path(const SceneView &S, const RenderView &R, const WfView &W, uint32_t slot, int depth, F3 tail,
                           Rng &rng, uint32_t sample) {
    F3 L = tail;
    for (int b = depth - 1; b >= 0; b--) {
        const float4 *e = wf_entry(W, slot, b);
        float4 e0 = e[0], e1 = e[1];
        L = f3(e0.x, e0.y, e0.z) + f3(e1.x, e1.y, e1.z) * L;
    }
    float4 *r = wf_rec(W, slot);
    float4 q3 = r[3];
    F3 accum = f3(q3.x, q3.y, q3.z) + L;
    sample++;
    int x, y; bool inside; size_t out_index;
    wf_slot_to_pixel(R, slot + W.slot_base, x, y, inside, out_index);
    if (sample < (uint32_t)R.samples) {
        F3 o, d;
        if (R.sample_seeds) wf_sample_seed(R, rng, slot + W.slot_base, x, y, sample);
        wf_camera_ray(S, R, rng, x, y, o, d);
        r[0] = make_float4(o.x, o.y, o.z, d.x);
        r[1] = make_float4(d.y, d.z, __uint_as_float(rng.x), rng.saved);
        r[3] = make_float4(accum.x, accum.y, accum.z, __uint_as_float(wf_pack(0, rng.has_saved, sample)));
        return sample < (uint32_t)R.sample_stop ? WF_NEXT_TRACE : WF_PARKED;
    }
    if (R.streams > 1) {                                             // throughput mode: this stream's unnormalised sum
        float *o = R.partial + 3 * (size_t)(slot + W.slot_base);
        o[0] = accum.x; o[1] = accum.y; o[2] = accum.z;
    } else {
        F3 px = R.inv_samples * accum;                               // scene.cpp:176
        if (R.out_rgb) { R.out_rgb[3 * out_index] = px.x; R.out_rgb[3 * out_index + 1] = px.y; R.out_rgb[3 * out_index + 2] = px.z; }
        if (R.out_rgb8) {                                            // sceneio.cpp:393-395
            R.out_rgb8[3 * out_index] = tonemap1(px.x); R.out_rgb8[3 * out_index + 1] = tonemap1(px.y); R.out_rgb8[3 * out_index + 2] = tonemap1(px.z);
        }
    }
    return 0;
}

// ---- init: seed every pixel, first camera ray, fill the round-0 trace queue ------------------------------
__global__ __launch_bounds__(256) void wf_init_kernel(SceneView S, RenderView R, WfView W) {
    __shared__ uint32_t buf[WF_BUF];
    __shared__ uint32_t cnt, gbase;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    Pusher q; q.buf = buf; q.cnt = &cnt;
    for (uint32_t base = blockIdx.x * 256u; base < W.n_slots; base += gridDim.x * 256u) {
        uint32_t slot = base + threadIdx.x;
        if (slot < W.n_slots) {
            int x, y; bool inside; size_t out_index;
            const uint32_t gslot = slot + W.slot_base;
            wf_slot_to_pixel(R, gslot, x, y, inside, out_index);
            if (!inside) { // padding of a border tile in the compact shard layout
                if (R.shard_count > 1 && (R.streams <= 1 || gslot < R.n_pixslots)) {
                    if (R.out_rgb) { R.out_rgb[3 * out_index] = 0.f; R.out_rgb[3 * out_index + 1] = 0.f; R.out_rgb[3 * out_index + 2] = 0.f; }
                    if (R.out_rgb8) { R.out_rgb8[3 * out_index] = 0; R.out_rgb8[3 * out_index + 1] = 0; R.out_rgb8[3 * out_index + 2] = 0; }
                }
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
                wf_push(q, slot);
            }
        }
        __syncthreads();
        if (cnt > WF_BUF - 256) wf_flush(q, W.q_trace[0], W.ctr + 0, &gbase);
    }
    __syncthreads();
    wf_flush(q, W.q_trace[0], W.ctr + 0, &gbase);
}

// ---- traversal kernels: persistent waves, per-lane LDS stacks ---------------------------------------------------
// Every wave owns a contiguous slice of the queue (no atomics: the work per ray is statistically uniform and a
// slice keeps neighbouring pixels together).  Lanes pull the next ray of the slice as soon as >= WF_REFILL lanes are
// idle.  Control flow is "while-while": lanes walk inner nodes until >= WF_LEAF_BATCH of them wait at a leaf,
// then the (expensive, division-heavy) triangle tests run for all waiting lanes together.
#define WF_REFILL 16
#define WF_LEAF_BATCH 20
#define WF_STEAL_CHUNK 64u
struct WfSlice { uint32_t pos, end, dyn_base, dyn_end, chunk; uint32_t *head; bool done; };
// The first (256 - share)/256 of a queue is cut into one static slice per wave of the population that owns the queue
// (blocks [first_block, first_block + n_blocks)); the rest is handed out in chunks through one atomic counter to waves
// whose slice has run dry — of either population — which evens out the tail (the slowest of ~4k static slices is ~20 %
// above the mean).  `owner = false` gives a wave of the OTHER population an empty slice that can only steal.
// dyn = chunk << 8 | share (packed by the host).
RT_DEV WfSlice wf_slice(uint32_t count, uint32_t *head, int dyn, uint32_t first_block, uint32_t n_blocks, bool owner) {
    uint32_t chunk = ((uint32_t)dyn >> 8) & 255u, share = (uint32_t)dyn & 255u;
    uint32_t nwaves = n_blocks * (blockDim.x >> 6);
    WfSlice s;
    s.head = head; s.chunk = chunk ? chunk : WF_STEAL_CHUNK; s.dyn_end = count;
    if (nwaves == 0) { s.pos = s.end = s.dyn_base = 0; s.done = count == 0; return s; } // nobody owns it: all of it is dynamic
    uint32_t stat = (uint32_t)(((unsigned long long)count * (256u - share)) >> 8);
    uint32_t per = (stat + nwaves - 1) / nwaves;
    per = (per + 63u) & ~63u;
    stat = share == 0 ? count : (per * nwaves < count ? per * nwaves : count);
    if (share == 0) { per = (count + nwaves - 1) / nwaves; per = (per + 63u) & ~63u; }
    s.dyn_base = stat;
    s.done = stat >= count;
    if (owner) {
        uint32_t wid = (blockIdx.x - first_block) * (blockDim.x >> 6) + (threadIdx.x >> 6);
        s.pos = wid * per < stat ? wid * per : stat;
        s.end = s.pos + per < stat ? s.pos + per : stat;
    } else s.pos = s.end = stat;
    return s;
}
// Wave-uniform: fetch the next dynamic chunk when the slice is empty.
RT_DEV void wf_steal(WfSlice &s) {
    uint32_t off = 0;
    if ((threadIdx.x & 63) == 0) off = atomicAdd(s.head, s.chunk);
    off = __shfl(off, 0);
    uint32_t b = s.dyn_base + off;
    if (b >= s.dyn_end) { s.done = true; return; }
    s.pos = b;
    s.end = b + s.chunk < s.dyn_end ? b + s.chunk : s.dyn_end;
}
// Hands queue positions to the lanes that want one; returns true for lanes that got `item`.
RT_DEV bool wf_take(WfSlice &s, bool want, uint32_t &item) {
    unsigned long long need = __ballot(want);
    uint32_t avail = s.end - s.pos;
    uint32_t rank = (uint32_t)__popcll(need & ((1ull << (threadIdx.x & 63)) - 1ull));
    bool got = want && rank < avail;
    if (got) item = s.pos + rank;
    uint32_t n = (uint32_t)__popcll(need);
    s.pos += n < avail ? n : avail;
    return got;
}

// Stack access.  The regular kernels index the LDS stack directly; the SPILL variant (selected on the host for trees deeper
// than WF_STACK) pays a bounds check per access and keeps the deep entries in a per-thread global area.
template <bool SPILL>
RT_DEV void wf_spush(uint32_t (*stack)[64], uint32_t *ovf, int lds_limit, int lane, int &sp, uint32_t v) {
    if (!SPILL || sp < lds_limit) stack[sp][lane] = v; else ovf[sp - lds_limit] = v;
    sp++;
}
template <bool SPILL>
RT_DEV uint32_t wf_spop(uint32_t (*stack)[64], const uint32_t *ovf, int lds_limit, int lane, int &sp) {
    --sp;
    return (!SPILL || sp < lds_limit) ? stack[sp][lane] : ovf[sp - lds_limit];
}

template <bool COUNT, bool SPILL>
RT_DEV void wf_trace_loop(const SceneView &S, const WfView &W, uint32_t (*stack)[64], const uint32_t *queue, WfSlice slice,
                          unsigned long long *counters, int refill, int leaf_batch, int lds_limit) {
    const int lane = threadIdx.x & 63;
    uint32_t *ovf = SPILL ? W.ovf + ((size_t)blockIdx.x * 256u + threadIdx.x) * WF_OVF : nullptr;
    bool active = false;
    uint32_t slot = 0, cur = 0, hit = WF_MISS;
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
    unsigned long long n_nodes = 0, n_tris = 0;
    unsigned long long w_node_iters = 0, w_leaf_phases = 0, w_leaf_lanes = 0, w_refills = 0; // wave-level (lane 0 reports)
    uint32_t w_iter = 0, ray_start = 0; // COUNT: wave iterations a ray stays in flight -> histogram counters[16 + min(15, iterations / 32)]
    const bool hist = COUNT && counters && counters[15] != 0; // the host sets counters[15] when the histogram is wanted (one global atomic per query)
    for (;;) {
        unsigned long long idle = __ballot(!active);
        if (idle && (slice.pos < slice.end || !slice.done) && (__popcll(idle) >= refill || idle == ~0ull)) {
            if (slice.pos >= slice.end) wf_steal(slice);
            uint32_t item = 0;
            if (COUNT) w_refills++;
            if (wf_take(slice, !active, item)) {
                slot = queue[item];
                const float4 *r = wf_rec(W, slot);
                float4 q0 = r[0], q1 = r[1];
                o = f3(q0.x, q0.y, q0.z); d = f3(q0.w, q1.x, q1.y);
                ray = make_ray_inv(o, d);
                h_ray = S.exact_boxes ? pt_look_behind_abs(d, S.box_c2x) : 0.f;
                cur = 0; sp = 0; hit = WF_MISS; best_t = RT_T_MAX; cull_t = RT_T_MAX; t2 = 2.f * RT_T_MAX; best_u = 0.f; best_v = 0.f;
                active = true;
                if (COUNT) ray_start = w_iter;
            }
        }
        const unsigned long long m_active = __ballot(active);
        if (!m_active) break;
        // A thinly populated wave (the drain of a launch, small frames) does not make leaf lanes wait for 20 companions.
        const int lb = min(leaf_batch & 255, (__popcll(m_active) * (leaf_batch >> 16) + 255) >> 8); // a share of the active lanes, in 1/256
        // phase 1: inner nodes
        for (;;) {
            bool inner = active && !(cur & RT_LEAF_BIT);
            unsigned long long m_inner = __ballot(inner);
            unsigned long long m_leaf = __ballot(active && (cur & RT_LEAF_BIT));
            if (!m_inner || __popcll(m_leaf) >= lb) break;
            if (COUNT) { w_node_iters++; w_iter++; }
            if (inner) {
                const float4 *q = reinterpret_cast<const float4 *>(S.nodes + cur);
                float4 lo0 = q[0], hi0 = q[1], lo1 = q[2], hi1 = q[3];
                if (COUNT) n_nodes++;
                float n0, n1;
                bool h0 = slab_test(lo0, hi0, ray, cull_t, n0);
                bool h1 = slab_test(lo1, hi1, ray, cull_t, n1);
                uint32_t c0 = __float_as_uint(lo0.w), c1 = __float_as_uint(lo1.w);
                if (h0 & h1) {
                    bool swap = n1 < n0;
                    wf_spush<SPILL>(stack, ovf, lds_limit, lane, sp, swap ? c0 : c1);
                    cur = swap ? c1 : c0;
                } else if (h0) cur = c0;
                else if (h1) cur = c1;
                else if (sp == 0) { // traversal finished: publish the hit
                    store_hit();
                    active = false;
                    if (COUNT && hist) { uint32_t b = (w_iter - ray_start) >> 5; atomicAdd(&counters[16 + (b > 15u ? 15u : b)], 1ull); }
                } else cur = wf_spop<SPILL>(stack, ovf, lds_limit, lane, sp);
            }
        }
        if (COUNT) w_iter++;
        // phase 2: every lane waiting at a leaf tests its triangles
        if (COUNT) { unsigned long long ml = __ballot(active && (cur & RT_LEAF_BIT)); if (ml) { w_leaf_phases++; w_leaf_lanes += __popcll(ml); } }
        if (active && (cur & RT_LEAF_BIT)) {
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
                active = false;
                if (COUNT && hist) { uint32_t b = (w_iter - ray_start) >> 5; atomicAdd(&counters[16 + (b > 15u ? 15u : b)], 1ull); }
            } else cur = wf_spop<SPILL>(stack, ovf, lds_limit, lane, sp);
        }
    }
    if (COUNT && counters) {
        atomicAdd(&counters[2], n_nodes); atomicAdd(&counters[3], n_tris);
        atomicAdd(&counters[8], n_nodes); atomicAdd(&counters[9], n_tris);
        if (lane == 0) { atomicAdd(&counters[4], w_node_iters); atomicAdd(&counters[5], w_leaf_phases); atomicAdd(&counters[6], w_leaf_lanes); atomicAdd(&counters[7], w_refills); }
    }
}

// All-hits light sum (FiguresMix::getTotalPdf, distributions.h:148-165) with the reference's addition tree.
// The finished sum goes straight into the pending bounce's pdf: E0.w += sum / n_lights (distributions.h:123,273).
template <bool COUNT, bool SPILL>
RT_DEV void wf_light_loop(const SceneView &S, const WfView &W, uint32_t (*stack)[64], const uint32_t *queue, WfSlice slice,
                          unsigned long long *counters, int refill, int leaf_batch, int lds_limit) {
    const int lane = threadIdx.x & 63;
    uint32_t *ovf = SPILL ? W.ovf + ((size_t)blockIdx.x * 256u + threadIdx.x) * WF_OVF : nullptr;
    bool active = false, descending = true;
    uint32_t slot = 0, cur = 0;
    int sp = 0;
    unsigned long long addmask = 0;
    float v = 0.f;
    F3 o = f3(0.f, 0.f, 0.f), d = f3(0.f, 0.f, 1.f);
    RayInv ray = make_ray_inv(o, d);
    unsigned long long n_nodes = 0, n_tris = 0;
    for (;;) {
        unsigned long long idle = __ballot(!active);
        if (idle && (slice.pos < slice.end || !slice.done) && (__popcll(idle) >= refill || idle == ~0ull)) {
            if (slice.pos >= slice.end) wf_steal(slice);
            uint32_t item = 0;
            if (wf_take(slice, !active, item)) {
                slot = queue[item];
                const float4 *r = wf_rec(W, slot);
                float4 q0 = r[0], q1 = r[1];
                o = f3(q0.x, q0.y, q0.z); d = f3(q0.w, q1.x, q1.y);
                ray = make_ray_inv(o, d);
                cur = 0; sp = 0; addmask = 0; v = 0.f; descending = true;
                active = true;
            }
        }
        const unsigned long long m_active = __ballot(active);
        if (!m_active) break;
        const int lb = min(leaf_batch & 255, (__popcll(m_active) * (leaf_batch >> 16) + 255) >> 8);
        // phase 1: node steps and (cheap) return steps, until enough lanes wait at a leaf
        for (;;) {
            bool at_leaf = active && descending && (cur & RT_LEAF_BIT);
            bool busy = active && !at_leaf;
            if (!__ballot(busy) || __popcll(__ballot(at_leaf)) >= lb) break;
            if (busy) {
                if (descending) {
                    const float4 *q = reinterpret_cast<const float4 *>(S.light_nodes + cur);
                    float4 lo0 = q[0], hi0 = q[1], lo1 = q[2], hi1 = q[3];
                    if (COUNT) n_nodes++;
                    float n0, n1;
                    bool h0 = slab_test(lo0, hi0, ray, RT_T_MAX, n0);
                    bool h1 = slab_test(lo1, hi1, ray, RT_T_MAX, n1);
                    uint32_t c0 = __float_as_uint(lo0.w), c1 = __float_as_uint(lo1.w);
                    if (h0 & h1) { addmask &= ~(1ull << sp); wf_spush<SPILL>(stack, ovf, lds_limit, lane, sp, c1); cur = c0; }
                    else if (h0) cur = c0;
                    else if (h1) cur = c1;
                    else { v = 0.f; descending = false; }
                } else if (sp == 0) { // sum complete
                    int depth = (int)(__float_as_uint(reinterpret_cast<const float *>(wf_rec(W, slot) + 3)[3]) & 15u);
                    float *pdf = reinterpret_cast<float *>(wf_entry(W, slot, depth)) + 3;
                    *pdf = *pdf + v / S.n_lights_f;
                    active = false;
                } else {
                    uint32_t f = wf_spop<SPILL>(stack, ovf, lds_limit, lane, sp);
                    if ((addmask >> sp) & 1ull) v = __uint_as_float(f) + v;       // left total + right total
                    else { addmask |= 1ull << sp; wf_spush<SPILL>(stack, ovf, lds_limit, lane, sp, __float_as_uint(v)); cur = f; descending = true; }
                }
            }
        }
        // phase 2: leaves
        if (active && descending && (cur & RT_LEAF_BIT)) {
            float result = 0.f;
            if (cur != RT_EMPTY_LEAF) {
                uint32_t i = cur & ~RT_LEAF_BIT;
                for (;;) {
                    bool last;
                    if (COUNT) n_tris++;
                    result += light_pdf_one(S.lights + i, o, d, last, S.hw7 != 0);
                    if (last) break;
                    i++;
                }
            }
            v = result;
            descending = false;
        }
    }
    if (COUNT && counters) { atomicAdd(&counters[2], n_nodes); atomicAdd(&counters[3], n_tris); }
}

// The same sum, decoupled: WHICH lights the ray hits is found by a plain all-hits walk (no frames, left child first, so the
// hits come out in the reference's light order); in WHAT ORDER their terms are added matters only for three or more hits
// (x + 0 = x, a + b = b + a) and is then rebuilt from the depths at which neighbouring hits separate in the reference tree
// (SceneView::light_sep): sum(node) = sum(left) + sum(right), a side without hits is the additive identity, a leaf adds its
// hits in index order.
// Hits are kept in the top of the lane's LDS stack column ({index, term} pairs growing downwards); a query whose hits would
// run into its stack, or with more than WF_MAX_LIGHT_HITS of them, is handed to wf_light_exact_kernel (frame walk of rt_device.h).
#define WF_MAX_LIGHT_HITS 5
template <bool COUNT>
RT_DEV void wf_light_loop_lean(const SceneView &S, const WfView &W, uint32_t (*stack)[64], const uint32_t *queue, WfSlice slice,
                               unsigned long long *counters, int refill, int leaf_batch, uint32_t *slow_count) {
    const int lane = threadIdx.x & 63;
    bool active = false, overflow = false;
    uint32_t slot = 0, cur = 0;
    int sp = 0, k = 0;
    F3 o = f3(0.f, 0.f, 0.f), d = f3(0.f, 0.f, 1.f);
    RayInv ray = make_ray_inv(o, d);
    unsigned long long n_nodes = 0, n_tris = 0;
    uint32_t w_iter = 0, ray_start = 0; // COUNT: histogram of in-flight wave iterations per query, counters[32 + ...]
    const bool hist = COUNT && counters && counters[15] != 0;
    // Hit j of the finished walk: light index in stack[31-2j], term in stack[30-2j] (ascending indices); the bottom of the
    // column is free by then and holds the separation depths while the terms are merged.
    auto finish = [&]() {
        active = false;
        if (overflow) { W.q_slow[atomicAdd(slow_count, 1u)] = slot; return; }
        float v = 0.f;
        if (k == 1) v = __uint_as_float(stack[WF_STACK - 2][lane]);
        else if (k == 2) v = __uint_as_float(stack[WF_STACK - 2][lane]) + __uint_as_float(stack[WF_STACK - 4][lane]);
        else if (k > 2) {
            // Separation depth of each pair of neighbouring hits (two independent table reads each), then merge the pair that
            // separates DEEPEST first: the node where they separate has exactly these two groups under its left and right
            // child, so this rebuilds sum(node) = sum(left) + sum(right) bottom-up; inside a leaf the pseudo depths make it
            // ((a + b) + c).  Terms live in column words 30-2j, depths in words 0..k-2.
            const uint32_t nl = S.n_lights;
            for (int j = 1; j < k; j++) {
                uint32_t a0 = stack[WF_STACK - 1 - 2 * (j - 1)][lane], b0 = stack[WF_STACK - 1 - 2 * j][lane]; // boundaries a0 .. b0-1
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
        *pdf = *pdf + v / S.n_lights_f;                                  // distributions.h:123,273
    };
    for (;;) {
        unsigned long long idle = __ballot(!active);
        if (idle && (slice.pos < slice.end || !slice.done) && (__popcll(idle) >= refill || idle == ~0ull)) {
            if (slice.pos >= slice.end) wf_steal(slice);
            uint32_t item = 0;
            if (wf_take(slice, !active, item)) {
                slot = queue[item];
                const float4 *r = wf_rec(W, slot);
                float4 q0 = r[0], q1 = r[1];
                o = f3(q0.x, q0.y, q0.z); d = f3(q0.w, q1.x, q1.y);
                ray = make_ray_inv(o, d);
                cur = 0; sp = 0; k = 0; overflow = false;
                active = true;
                if (COUNT) ray_start = w_iter;
            }
        }
        const unsigned long long m_active = __ballot(active);
        if (!m_active) break;
        const int lb = min(leaf_batch & 255, (__popcll(m_active) * (leaf_batch >> 16) + 255) >> 8); // a share of the active lanes, in 1/256
        for (;;) { // phase 1: inner nodes
            bool inner = active && !(cur & RT_LEAF_BIT);
            if (!__ballot(inner) || __popcll(__ballot(active && (cur & RT_LEAF_BIT))) >= lb) break;
            if (COUNT) w_iter++;
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
                else if (sp == 0) { finish(); if (COUNT && hist) { uint32_t b = (w_iter - ray_start) >> 5; atomicAdd(&counters[32 + (b > 15u ? 15u : b)], 1ull); } }
                else cur = stack[--sp][lane];
            }
        }
        if (COUNT) w_iter++;
        if (active && (cur & RT_LEAF_BIT)) { // phase 2: leaves
            if (cur != RT_EMPTY_LEAF) {
                uint32_t i = cur & ~RT_LEAF_BIT;
                for (;;) {
                    bool last, robust; uint32_t li;
                    if (COUNT) n_tris++;
                    float term = pt_light_pdf_one(S, S.lights + i, o, d, last, robust, li); // the reference topology over `lights`: the position i IS the light index (li is not used)
                    if (term != 0.f) { // a hit (a miss contributes exactly 0, and adding 0 changes nothing)
                        if (!robust || k >= WF_MAX_LIGHT_HITS || sp + 2 * k + 2 >= WF_STACK) overflow = true; // not robust against the reference's box tests: exact walk
                        else { stack[WF_STACK - 1 - 2 * k][lane] = i; stack[WF_STACK - 2 - 2 * k][lane] = __float_as_uint(term); k++; }
                    }
                    if (last) break;
                    i++;
                }
            }
            if (sp == 0) { finish(); if (COUNT && hist) { uint32_t b = (w_iter - ray_start) >> 5; atomicAdd(&counters[32 + (b > 15u ? 15u : b)], 1ull); } }
            else cur = stack[--sp][lane];
        }
    }
    if (COUNT && counters) { atomicAdd(&counters[2], n_nodes); atomicAdd(&counters[3], n_tris); }
}

// ---- traverse: both loops in one persistent launch -------------------------------------------------------------------
// Blocks [0, nb_t) own the trace queue, the rest own the light queue; nb_t follows the queue lengths weighted by the
// measured cost of one query of each kind (about equal on the benchmark scene since the light loop became frame-free).
// After its own queue a block helps with the other one's dynamic tail, so a wrong split only costs a few chunks.
template <bool COUNT, bool SPILL>
__global__ __launch_bounds__(256) void wf_traverse_kernel(SceneView S, WfView W, uint32_t round, unsigned long long *counters,
                                                          int t_refill, int t_batch, int l_refill, int l_batch, int dyn, int lds_limit) {
    __shared__ uint32_t lds_stack[4][WF_STACK][64];
    uint32_t(*stack)[64] = lds_stack[threadIdx.x >> 6];
    const uint32_t ct = W.ctr[WF_CTR * round + 0], cl = S.n_lights ? W.ctr[WF_CTR * round + 1] : 0u;
    const uint32_t nb = gridDim.x;
    uint32_t nb_t = nb;
    if (cl) {
        unsigned long long kt = ((uint32_t)dyn >> 16) & 255u, kl = (uint32_t)dyn >> 24;  // cost weights packed by the host (0 = default 1 : 1, measured best on the benchmark scene)
        if (!kt || !kl) { kt = 1; kl = 1; }
        unsigned long long wt = kt * ct, wl = kl * cl;
        nb_t = (uint32_t)((wt * nb + (wt + wl) / 2) / (wt + wl));
        if (nb_t < 1u) nb_t = 1u;
        if (nb_t > nb - 1u) nb_t = nb - 1u;
    }
    const uint32_t *q_t = W.q_trace[round & 1], *q_l = W.q_light;
    uint32_t *head_t = W.ctr + WF_CTR * round + 2, *head_l = W.ctr + WF_CTR * round + 3;
    const bool tracer = blockIdx.x < nb_t;
    if (tracer) wf_trace_loop<COUNT, SPILL>(S, W, stack, q_t, wf_slice(ct, head_t, dyn, 0u, nb_t, true), counters, t_refill, t_batch, lds_limit);
    if (cl) {
        if (SPILL) wf_light_loop<COUNT, true>(S, W, stack, q_l, wf_slice(cl, head_l, dyn, nb_t, nb - nb_t, !tracer), counters, l_refill, l_batch, lds_limit);
        else wf_light_loop_lean<COUNT>(S, W, stack, q_l, wf_slice(cl, head_l, dyn, nb_t, nb - nb_t, !tracer), counters, l_refill, l_batch, W.ctr + WF_CTR * round + 4);
    }
    if (!tracer) wf_trace_loop<COUNT, SPILL>(S, W, stack, q_t, wf_slice(ct, head_t, dyn, 0u, nb_t, false), counters, t_refill, t_batch, lds_limit);
}

// The few light queries the lean loop could not finish: the plain reference-order frame walk (light_pdf_sum), one lane each.
__global__ __launch_bounds__(64) void wf_light_exact_kernel(SceneView S, WfView W, uint32_t round) {
    const uint32_t count = W.ctr[WF_CTR * round + 4];
    uint32_t stack[RT_STACK_SIZE];
    for (uint32_t i = blockIdx.x * 64u + threadIdx.x; i < count; i += gridDim.x * 64u) {
        uint32_t slot = W.q_slow[i];
        const float4 *r = wf_rec(W, slot);
        float4 q0 = r[0], q1 = r[1];
        Counters cnt; cnt.closest = cnt.lightq = cnt.nodes = cnt.tris = 0;
        const F3 x = f3(q0.x, q0.y, q0.z), dir = f3(q0.w, q1.x, q1.y);
        float v = S.exact_boxes ? ref_light_pdf_sum(S, x, dir, stack) : light_pdf_sum<false>(S, x, dir, stack, cnt);
        int depth = (int)(__float_as_uint(r[3].w) & 15u);
        float *pdf = reinterpret_cast<float *>(wf_entry(W, slot, depth)) + 3;
        *pdf = *pdf + v / S.n_lights_f;
    }
}

// ---- shade: finish the pending bounce (scene.cpp:158-164), then scene.cpp:89-156 for the new hit ------------------------
// Returns what the slot needs next: WF_NEXT_TRACE (its record holds a ray to trace: a new bounce or the next sample's camera
// ray), | WF_NEXT_LIGHT (that ray is also a bounce's light-pdf query), WF_PARKED (the next sample's camera ray is in the
// record but belongs to the frame's next phase), or 0 (the pixel is finished and written).
// FEAT: which optional features the shading code is compiled with (WF_FEAT_ENV: environment-map lookup on a miss; WF_FEAT_HW7: the
// hw7 material model).  A kernel variant without a feature the render cannot use carries less register pressure: the inlined
// double-precision atan2 / asin of the environment lookup alone double the spills of the persistent kernel.
#define WF_FEAT_ENV 1
#define WF_FEAT_HW7 2
template <int FEAT = WF_FEAT_ENV | WF_FEAT_HW7>
RT_DEV int wf_shade_item(const SceneView &S, const RenderView &R, const WfView &W, uint32_t slot, unsigned long long *counters, bool *discarded = nullptr) {
    float4 *r = wf_rec(W, slot);
    float4 q0 = r[0], q1 = r[1], q2 = r[2];
    F3 d = f3(q0.w, q1.x, q1.y);
    Rng rng; rng.x = __float_as_uint(q1.z); rng.saved = q1.w;
    const uint32_t packed = __float_as_uint(reinterpret_cast<const float *>(r + 3)[3]);
    int depth = (int)(packed & 15u);
    rng.has_saved = (packed & 16u) != 0;
    const uint32_t sample = (packed >> 6) & WF_SAMPLE_MASK;
    // A path that ends here (miss, clamp, early-out, deepest level) returns `tail` from its innermost call below `levels` bounces.
    bool ended = false;
    F3 tail = f3(0.f, 0.f, 0.f);
    int levels = 0;
    if (packed & WF_PENDING_BIT) {
        // The bounce at `depth` sampled the ray that was just traced; its pdf is complete now (Mix::pdf, distributions.h:268-278).
        float4 *e = wf_entry(W, slot, depth);
        float4 e0 = e[0], e1 = e[1];
        F3 brdf = f3(e1.x, e1.y, e1.z);
        float pdf = e0.w / S.n_components_f;                              // :278
        float k = (float)(1. / (double)pdf * fabs((double)e1.w));              // scene.cpp:159
        F3 mult = k * brdf;
        bool clamp = mult.x > 6.f || mult.y > 6.f || mult.z > 6.f || mult.x != mult.x || mult.y != mult.y || mult.z != mult.z;
        if (clamp || depth + 1 >= R.ray_depth) {
            // clamp hack (scene.cpp:161-163): the path returns the emission and the speculative hit is dropped; at the
            // last level the inner call returns 0, i.e. emission + mult * 0 evaluated literally.
            if (counters) atomicAdd(&counters[10], 1ull);
            if (discarded) *discarded = true;
            ended = true;
            tail = f3(e0.x, e0.y, e0.z);
            levels = depth;
            if (!clamp) { e[1] = make_float4(mult.x, mult.y, mult.z, e1.w); tail = f3(0.f, 0.f, 0.f); levels = depth + 1; }
        } else {
            bool survives = true;
            if (R.rr_depth > 0 && depth + 1 >= R.rr_depth) {
                // Russian roulette (throughput mode only): the bounce continues with probability q and carries mult / q, else the
                // path returns its emission as if the inner call were 0 (the traced hit is dropped like a clamped one).
                const float q = wf_roulette_q(W, slot, depth, mult);
                survives = rng_u01(rng) < q;
                mult = (1.f / q) * mult;
            }
            e[1] = make_float4(mult.x, mult.y, mult.z, e1.w);
            if (survives) depth++;
            else {
                if (counters) atomicAdd(&counters[10], 1ull);
                if (discarded) *discarded = true;
                ended = true; tail = f3(0.f, 0.f, 0.f); levels = depth + 1;
            }
        }
    }
    if (!ended) {
        const uint32_t hit = __float_as_uint(q2.w);
        levels = depth;
        if (hit == WF_MISS) { ended = true; tail = miss_color<(FEAT & WF_FEAT_ENV) != 0>(S, d); }
        else {
            HitRec h;
            h.idx = (int)(hit & WF_INDEX_MASK); h.inside = (hit & WF_INSIDE_BIT) != 0; h.t = q2.x; h.u = q2.y; h.v = q2.z;
            if (S.last_level_emission_only && depth + 1 >= R.ray_depth) {
                // Deepest level: whatever Mix::sample / brdf / pdf produce, getColor returns its emission (see
                // SceneView::last_level_emission_only).  Only the random draws must still happen, in order
                // (distributions.h:257, then 3 normals | u1,u2 | index,u,v).
                tail = emission_fetch(S, h);
                int comp = (int)(rng_u01(rng) * S.n_components_f);
                if (comp == 0) { rng_n01(rng); rng_n01(rng); rng_n01(rng); }
                else if (comp == 2) { rng_u01(rng); rng_u01(rng); rng_u01(rng); }
                else { rng_u01(rng); rng_u01(rng); }
                ended = true;
            } else {
                const bool hw7 = (FEAT & WF_FEAT_HW7) && S.hw7;
                float4 *e = wf_entry(W, slot, depth);
                F3 bc; float metallic_eff, alpha; F3 sn;
                {
                    // Everything the sampling code below does not need leaves the registers here: the emission and the next ray's
                    // origin go to the path's record at once (their final places), colour and metallic shrink to the products
                    // the BRDF uses.  (hw7: colour texture = 1 and texture metallic = 1, so the products are the plain factors.)
                    F3 ng, base_color; float base_metallic; Shaded sh;
                    shade_fetch(S, h, ng, sh, base_color, base_metallic, hw7);
                    F3 x = f3(q0.x, q0.y, q0.z) + h.t * d;                             // scene.cpp:104
                    F3 xo = x + 9.99999974737875163555e-05f * ng;                      // x + eps * geomNorma
                    e[0] = make_float4(sh.emission.x, sh.emission.y, sh.emission.z, 0.f);
                    r[0] = make_float4(xo.x, xo.y, xo.z, 0.f);                        // the next ray doubles as the light query
                    bc = base_color * sh.color;
                    metallic_eff = sh.metallic * base_metallic;
                    alpha = sh.alpha; sn = sh.sn;
                }
                int comp = (int)(rng_u01(rng) * S.n_components_f);               // distributions.h:257
                F3 nd;
                if (comp == 0) nd = cosine_sample(rng, sn);
                else if (comp == 2) { const float4 xq = r[0]; nd = light_sample(S, rng, f3(xq.x, xq.y, xq.z)); }
                else nd = vndf_sample(rng, sn, d, alpha);
                F3 brdf = hw7 ? material_brdf_hw7(bc, metallic_eff, nd, neg(d), sn, alpha * alpha)
                              : material_brdf_pre(bc, metallic_eff, nd, neg(d), sn, alpha);
                const float epsf = 9.99999974737875163555e-05f;
                if (brdf.x <= epsf && brdf.y <= epsf && brdf.z <= epsf) {             // scene.cpp:154-156
                    const float4 e0 = e[0];
                    ended = true; tail = f3(e0.x, e0.y, e0.z);
                } else {
                    float pdf = 0.f;                                                   // distributions.h:268-276, first two terms
                    pdf += cosine_pdf(sn, nd);
                    pdf += vndf_pdf(sn, nd, d, alpha);
                    reinterpret_cast<float *>(e)[3] = pdf;
                    e[1] = make_float4(brdf.x, brdf.y, brdf.z, dot(nd, sn));
                    reinterpret_cast<float *>(r)[3] = nd.x;
                    r[1] = make_float4(nd.y, nd.z, __uint_as_float(rng.x), rng.saved);
                    reinterpret_cast<float *>(r + 3)[3] = __uint_as_float(wf_pack(depth, rng.has_saved, sample, true));
                    return WF_NEXT_TRACE | (S.n_lights ? WF_NEXT_LIGHT : 0);           // traced speculatively beside its own light-pdf sum
                }
            }
        }
    }
    return wf_finish_path(S, R, W, slot, levels, tail, rng, sample);
}

// wf_shade_item behind the exactness gate (rt_exact.h, pt_hit_stands): a hit that does not stand as the reference's answer
// and has not been through the exact walk yet goes there first (PT_SHADE_EXACT: nothing of the path's state is touched).
template <int FEAT = WF_FEAT_ENV | WF_FEAT_HW7>
RT_DEV int pt_shade_item(const SceneView &S, const RenderView &R, const WfView &W, uint32_t slot, bool &discarded, unsigned long long *counters = nullptr) {
    if (S.exact_boxes) {
        const float4 *r = wf_rec(W, slot);
        const float4 q2 = r[2];
        const uint32_t hit = __float_as_uint(q2.w);
        const uint32_t packed = __float_as_uint(reinterpret_cast<const float *>(r + 3)[3]);
        if (S.n_tripwire_groups && !(packed & WF_VERIFIED_BIT)) { // the ray crossed a tripwire (rt_exact.h pt_tripwire): hit or miss, the exact walk decides
            const float4 q0 = r[0], q1 = r[1];
            if (pt_tripwire(S, f3(q0.x, q0.y, q0.z), f3(q0.w, q1.x, q1.y))) return PT_SHADE_EXACT;
        }
        if (hit != WF_MISS && !(packed & WF_VERIFIED_BIT)) {
            const float4 q0 = r[0], q1 = r[1];
            const F3 o = f3(q0.x, q0.y, q0.z), d = f3(q0.w, q1.x, q1.y);
            const float4 *bx = reinterpret_cast<const float4 *>(S.tri_box) + 2 * (size_t)(hit & WF_INDEX_MASK);
            const float4 lo = bx[0], hi = bx[1];
            if (!pt_hit_stands(f3(lo.x, lo.y, lo.z), f3(hi.x, hi.y, hi.z), o, d, q2.x, pt_gap_floor(hit, q2.x), S.box_c2, S.box_c2x, S.cull_k)) return PT_SHADE_EXACT;
        }
    }
    return wf_shade_item<FEAT>(S, R, W, slot, counters, &discarded);
}

// Throughput mode epilogue: pixel = float(1/spp) * (sum of its K stream sums, added in stream order), then the usual tonemap.
__global__ __launch_bounds__(256) void wf_reduce_streams_kernel(RenderView R) {
    for (uint32_t p = blockIdx.x * 256u + threadIdx.x; p < R.n_pixslots; p += gridDim.x * 256u) {
        int x, y; bool inside; size_t out_index;
        slot_to_pixel(R, p, x, y, inside, out_index);
        if (!inside) continue;
        F3 sum = f3(0.f, 0.f, 0.f);
        for (int k = 0; k < R.streams; k++) {
            const float *q = R.partial + 3 * ((size_t)k * R.n_pixslots + p);
            sum = sum + f3(q[0], q[1], q[2]);
        }
        F3 px = R.inv_samples * sum;
        if (R.out_rgb) { R.out_rgb[3 * out_index] = px.x; R.out_rgb[3 * out_index + 1] = px.y; R.out_rgb[3 * out_index + 2] = px.z; }
        if (R.out_rgb8) { R.out_rgb8[3 * out_index] = tonemap1(px.x); R.out_rgb8[3 * out_index + 1] = tonemap1(px.y); R.out_rgb8[3 * out_index + 2] = tonemap1(px.z); }
    }
}

#ifndef WF_SHADE_OCC
#define WF_SHADE_OCC 3          // measured: 3 resident blocks per CU (some spills) beats the unconstrained 2 by ~1.5 %
#endif
__global__ __launch_bounds__(256, WF_SHADE_OCC) void wf_shade_kernel(SceneView S, RenderView R, WfView W, uint32_t round, unsigned long long *counters) {
    __shared__ uint32_t buf_l[WF_BUF], buf_n[WF_BUF];
    __shared__ uint32_t cnt_l, cnt_n, gbase;
    if (threadIdx.x == 0) { cnt_l = 0; cnt_n = 0; }
    __syncthreads();
    Pusher to_light; to_light.buf = buf_l; to_light.cnt = &cnt_l;
    Pusher next; next.buf = buf_n; next.cnt = &cnt_n;
    const uint32_t *queue = W.q_trace[round & 1];
    const uint32_t count = W.ctr[WF_CTR * round + 0];
    uint32_t *next_queue = W.q_trace[(round + 1) & 1], *next_count = W.ctr + WF_CTR * (round + 1) + 0, *light_count = W.ctr + WF_CTR * (round + 1) + 1;
    for (uint32_t base = blockIdx.x * 256u; base < count; base += gridDim.x * 256u) {
        uint32_t i = base + threadIdx.x;
        if (i < count) {
            const uint32_t slot = queue[i];
            bool discarded = false;
            const int todo = pt_shade_item(S, R, W, slot, discarded, counters);
            if (todo == PT_SHADE_EXACT) W.q_slow[atomicAdd(W.ctr + WF_CTR * round + 5, 1u)] = slot; // rare (~1e-5): wf_trace_exact_kernel takes it from here
            else {
                if (todo & WF_NEXT_TRACE) wf_push(next, slot);
                if (todo & WF_NEXT_LIGHT) wf_push(to_light, slot);
            }
        }
        __syncthreads();
        if (cnt_l > WF_BUF - 256) wf_flush(to_light, W.q_light, light_count, &gbase);
        if (cnt_n > WF_BUF - 256) wf_flush(next, next_queue, next_count, &gbase);
    }
    __syncthreads();
    wf_flush(to_light, W.q_light, light_count, &gbase);
    wf_flush(next, next_queue, next_count, &gbase);
}

// The few paths whose hit the shader would not take at face value (pt_shade_item): walk them again with the reference's own box
// arithmetic (ref_closest_hit), mark the hit verified, shade them and append what they need next to the next round's queues.
// One lane per path; a launch usually finds nothing to do (the queue holds ~1e-5 of the round's paths).
__global__ __launch_bounds__(64) void wf_trace_exact_kernel(SceneView S, RenderView R, WfView W, uint32_t round, unsigned long long *counters) {
    const uint32_t count = W.ctr[WF_CTR * round + 5];
    uint32_t stack[RT_STACK_SIZE];
    uint32_t *next_queue = W.q_trace[(round + 1) & 1], *next_count = W.ctr + WF_CTR * (round + 1) + 0, *light_count = W.ctr + WF_CTR * (round + 1) + 1;
    for (uint32_t i = blockIdx.x * 64u + threadIdx.x; i < count; i += gridDim.x * 64u) {
        const uint32_t slot = W.q_slow[i];
        float4 *r = wf_rec(W, slot);
        const float4 q0 = r[0], q1 = r[1];
        float bt, bu, bv; uint32_t hit;
        ref_closest_hit(S, f3(q0.x, q0.y, q0.z), f3(q0.w, q1.x, q1.y), stack, bt, bu, bv, hit);
        r[2] = make_float4(bt, bu, bv, __uint_as_float(hit));
        float *pk = reinterpret_cast<float *>(r + 3) + 3;
        *pk = __uint_as_float(__float_as_uint(*pk) | WF_VERIFIED_BIT);
        if (counters) atomicAdd(&counters[12], 1ull);
        const int todo = wf_shade_item(S, R, W, slot, counters);
        if (todo & WF_NEXT_TRACE) next_queue[atomicAdd(next_count, 1u)] = slot;
        if (todo & WF_NEXT_LIGHT) W.q_light[atomicAdd(light_count, 1u)] = slot;
    }
}

} // namespace dev
} // namespace rtamd
