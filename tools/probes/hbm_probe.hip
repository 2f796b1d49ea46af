// Micro-probe: what this MI355X sustains for dense streaming reads / writes / copies, by access shape - the ceiling the HBM-bound kernels
// (k_wgrad_ring8: 1 KiB LDS-DMA pieces; k_adam, k_wgrad_reduce: 16-byte loads) are priced against in DESIGN.md section 4.2.
// build: hipcc --offload-arch=gfx950 -O3 -o hbm_probe hbm_probe.hip ; run: ./hbm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

// grid-stride 16-byte loads, U in flight per thread
template <int U>
__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ p, size_t n4, float* out) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = p[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    if (s.x + s.y + s.z + s.w == 12345.f) out[0] = s.x;
}
// each workgroup walks its own contiguous chunk (the shape of a stash walk): 1 KiB per wave instruction
template <int U>
__global__ __launch_bounds__(256) void k_read_chunk(const float4* __restrict__ p, size_t n4, float* out) {
    const size_t per = n4 / gridDim.x;
    const float4* q = p + (size_t)blockIdx.x * per;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t i = threadIdx.x; i + (U - 1) * 256 < per; i += U * 256) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = q[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    if (s.x + s.y + s.z + s.w == 12345.f) out[0] = s.x;
}
__global__ __launch_bounds__(256) void k_write(float4* __restrict__ p, size_t n4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) b[i] = a[i];
}

template <typename F>
static void timeit(const char* name, double bytes, F launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("%-46s %8.3f ms  %6.2f TB/s\n", name, best, bytes / best * 1e-9);
}

int main() {
    const size_t bytes = (size_t)4 << 30;            // 4 GiB: well past the 256 MB memory-side cache
    const size_t n4 = bytes / 16;
    float4 *a, *b; float* out;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&out, 256);
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
    for (int wg : {1024, 2048, 8192}) {
        char nm[96];
        snprintf(nm, sizeof nm, "read  grid-stride x4 in flight, %d wgs", wg);
        timeit(nm, (double)bytes, [&] { hipLaunchKernelGGL(k_read<4>, dim3(wg), dim3(256), 0, 0, a, n4, out); });
        snprintf(nm, sizeof nm, "read  grid-stride x8 in flight, %d wgs", wg);
        timeit(nm, (double)bytes, [&] { hipLaunchKernelGGL(k_read<8>, dim3(wg), dim3(256), 0, 0, a, n4, out); });
        snprintf(nm, sizeof nm, "read  per-workgroup chunks x8, %d wgs", wg);
        timeit(nm, (double)bytes, [&] { hipLaunchKernelGGL(k_read_chunk<8>, dim3(wg), dim3(256), 0, 0, a, n4, out); });
        snprintf(nm, sizeof nm, "write grid-stride, %d wgs", wg);
        timeit(nm, (double)bytes, [&] { hipLaunchKernelGGL(k_write, dim3(wg), dim3(256), 0, 0, b, n4); });
        snprintf(nm, sizeof nm, "copy  grid-stride (read + write bytes), %d wgs", wg);
        timeit(nm, 2.0 * bytes, [&] { hipLaunchKernelGGL(k_copy, dim3(wg), dim3(256), 0, 0, a, b, n4); });
    }
    return 0;
}
