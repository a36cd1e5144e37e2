// Test-only entry points (NOT part of include/rtamd.h, built into librtamd_testhooks.so): run individual
// device functions of rt_device.h on the GPU so tests can compare them with the host libm / libstdc++.
#include <hip/hip_runtime.h>
#include "device/rt_device.h"

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

extern "C" {
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
