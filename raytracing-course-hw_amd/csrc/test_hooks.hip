// Test-only entry points (NOT part of include/rtamd.h, built into librtamd_testhooks.so): run individual
// device functions of rt_device.h on the GPU so tests can compare them with the host libm / libstdc++.
#include <hip/hip_runtime.h>
#include "device/rt_device.h"
#include "device/rt_exact.h"
#include "device/rt_node_grid.h"
#include "host/fold_nodes.h"
#include <cstring>
#include <vector>

using namespace rtamd::dev;

__global__ void k_logf(const float *in, float *out, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) out[i] = rt_logf(in[i]);
}
__global__ void k_rng(uint32_t seed0, int n_seeds, int n_u, int n_n, float *out) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seeds) return;
    Rng r;
    rng_seed(r, seed0 + (uint32_t)s);
    float *o = out + (size_t)s * (n_u + n_n);
    for (int i = 0; i < n_u; i++) o[i] = rng_u01(r);
    for (int i = 0; i < n_n; i++) o[n_u + i] = rng_n01(r);
}

// rt_exact.h: the runner-up's distance as it travels in the hit word, and the gate's decision on a box / ray / hit / gap
__global__ void k_gap(const float *t, const float *t2, float *floor_out, uint32_t *code_out, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = pt_gap_code(t[i], t2[i]);
    code_out[i] = c;
    floor_out[i] = pt_gap_floor(c | 5u, t[i]);   // some figure index in the low bits: it must not disturb the code
}
__global__ void k_stands(const float *in, uint32_t *out, size_t n) { // per case 16 floats: lo.xyz hi.xyz o.xyz d.xyz t gap c2 cull_k
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *p = in + 16 * i;
    out[i] = pt_hit_stands(f3(p[0], p[1], p[2]), f3(p[3], p[4], p[5]), f3(p[6], p[7], p[8]), f3(p[9], p[10], p[11]), p[12], p[13], p[14], 1.25f * p[14], p[15]) ? 1u : 0u;
}

// rt_device.h slab_test_q against slab_test: a box on the walkers' 16-bit grid must be entered by every ray that enters the float box
__global__ void k_slab_q(const float *in, rtamd::NodeGrid G, uint32_t *out, size_t n) { // per case 13 floats: lo.xyz hi.xyz o.xyz d.xyz tbest
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *p = in + 13 * i;
    const F3 o = f3(p[6], p[7], p[8]), d = f3(p[9], p[10], p[11]);
    float tn = 0.f, tq = 0.f;
    const bool hf = slab_test(make_float4(p[0], p[1], p[2], 0.f), make_float4(p[3], p[4], p[5], 0.f), make_ray_inv(o, d), p[12], tn);
    bool fits = true;
    uint4 b;
    b.x = rtamd::grid_axis_word(p[0], p[3], G.lo[0], G.step[0], fits);
    b.y = rtamd::grid_axis_word(p[1], p[4], G.lo[1], G.step[1], fits);
    b.z = rtamd::grid_axis_word(p[2], p[5], G.lo[2], G.step[2], fits);
    b.w = 0u;
    const bool hq = slab_test_q(b, make_ray_grid(G, o, d), p[12], tq);
    out[i] = (hf ? 1u : 0u) | (hq ? 2u : 0u) | (fits ? 4u : 0u) | (tq <= tn ? 8u : 0u);
}

extern "C" {
// host/fold_nodes.h on host memory (no GPU needed): nodes = n two-box nodes of 64 bytes, grid_box = lo.xyz hi.xyz of what the grid must hold,
// out = room for cap wide nodes of 64 bytes, grid_out = the grid's lo[3], step[3], istep[3].  Returns the number of wide nodes, -1 on error.
int rtt_fold_nodes(const void *nodes, uint32_t n, const float *grid_box, void *out, uint32_t cap, uint32_t *depth_out, float *grid_out) {
    try {
        std::vector<rtamd::GpuNode> in((const rtamd::GpuNode *)nodes, (const rtamd::GpuNode *)nodes + n);
        const rtamd::NodeGrid G = rtamd::make_node_grid(grid_box, grid_box + 3);
        std::vector<rtamd::GpuNode4Q> wide;
        uint32_t depth = 0;
        rtamd::fold_nodes(in, G, wide, depth);
        if (wide.size() > cap) return -1;
        memcpy(out, wide.data(), wide.size() * sizeof(rtamd::GpuNode4Q));
        *depth_out = depth;
        for (int k = 0; k < 3; k++) { grid_out[k] = G.lo[k]; grid_out[3 + k] = G.step[k]; grid_out[6 + k] = G.istep[k]; }
        return (int)wide.size();
    } catch (...) { return -1; }
}
// grid_box: lo.xyz hi.xyz of what the grid must hold.  out bits: 1 float box entered, 2 grid box entered, 4 the box fits the grid, 8 grid entry <= float entry
int rtt_slab_q(const float *cases, const float *grid_box, uint32_t *out, size_t n) {
    const rtamd::NodeGrid G = rtamd::make_node_grid(grid_box, grid_box + 3);
    float *d_in = nullptr; uint32_t *d_out = nullptr;
    if (hipMalloc((void **)&d_in, n * 13 * 4) != hipSuccess || hipMalloc((void **)&d_out, n * 4) != hipSuccess) return -1;
    int rc = -1;
    if (hipMemcpy(d_in, cases, n * 13 * 4, hipMemcpyHostToDevice) == hipSuccess) {
        hipLaunchKernelGGL(k_slab_q, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_in, G, d_out, n);
        if (hipDeviceSynchronize() == hipSuccess && hipMemcpy(out, d_out, n * 4, hipMemcpyDeviceToHost) == hipSuccess) rc = 0;
    }
    (void)hipFree(d_in); (void)hipFree(d_out);
    return rc;
}
int rtt_gap_code(const float *t, const float *t2, float *floor_out, uint32_t *code_out, size_t n) {
    float *d_t = nullptr, *d_t2 = nullptr, *d_f = nullptr; uint32_t *d_c = nullptr;
    if (hipMalloc((void **)&d_t, n * 4) != hipSuccess || hipMalloc((void **)&d_t2, n * 4) != hipSuccess || hipMalloc((void **)&d_f, n * 4) != hipSuccess || hipMalloc((void **)&d_c, n * 4) != hipSuccess) return -1;
    (void)hipMemcpy(d_t, t, n * 4, hipMemcpyHostToDevice); (void)hipMemcpy(d_t2, t2, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_gap, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_t, d_t2, d_f, d_c, n);
    int rc = (hipMemcpy(floor_out, d_f, n * 4, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(code_out, d_c, n * 4, hipMemcpyDeviceToHost) == hipSuccess) ? 0 : -2;
    (void)hipFree(d_t); (void)hipFree(d_t2); (void)hipFree(d_f); (void)hipFree(d_c);
    return rc;
}
int rtt_hit_stands(const float *cases16, uint32_t *out, size_t n) {
    float *d_in = nullptr; uint32_t *d_out = nullptr;
    if (hipMalloc((void **)&d_in, n * 64) != hipSuccess || hipMalloc((void **)&d_out, n * 4) != hipSuccess) return -1;
    (void)hipMemcpy(d_in, cases16, n * 64, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_stands, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_in, d_out, n);
    int rc = hipMemcpy(out, d_out, n * 4, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
    (void)hipFree(d_in); (void)hipFree(d_out);
    return rc;
}
int rtt_logf(const float *in, float *out, size_t n) {
    float *d_in = nullptr, *d_out = nullptr;
    if (hipMalloc((void **)&d_in, n * 4) != hipSuccess || hipMalloc((void **)&d_out, n * 4) != hipSuccess) return -1;
    if (hipMemcpy(d_in, in, n * 4, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d_in); (void)hipFree(d_out); return -2; }
    hipLaunchKernelGGL(k_logf, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_in, d_out, n);
    int rc = hipMemcpy(out, d_out, n * 4, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
    (void)hipFree(d_in); (void)hipFree(d_out);
    return rc;
}
// streams for seeds seed0 .. seed0+n_seeds-1: n_u uniforms then n_n normals each
int rtt_rng_streams(uint32_t seed0, int n_seeds, int n_u, int n_n, float *out) {
    float *d_out = nullptr;
    size_t n = (size_t)n_seeds * (n_u + n_n);
    if (hipMalloc((void **)&d_out, n * 4) != hipSuccess) return -1;
    hipLaunchKernelGGL(k_rng, dim3((n_seeds + 63) / 64), dim3(64), 0, 0, seed0, n_seeds, n_u, n_n, d_out);
    int rc = hipMemcpy(out, d_out, n * 4, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
    (void)hipFree(d_out);
    return rc;
}
}
