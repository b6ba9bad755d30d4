// Microbenchmark (round 6): does the random-line rate out of the shipped 6 MB table (172 G lines/s against 281 out of 0.5 MB,
// tools/line_rate_scaling_bench.hip) come back when every XCD only ever touches ITS slice of the table?  Each XCD has its own 4 MB L2; workgroups
// are dealt round-robin over the 8 XCDs (blockIdx % 8).  Modes: whole = every workgroup draws lines from the whole region (today's forward: every wave
// gathers all 16 levels); part G = the region is cut into G slices and a workgroup draws from slice (blockIdx % 8) * G / 8 (a forward whose workgroups
// gather only the levels of their XCD's group).  Also whole regions of 0.5 ... 6 MB: where the rate drops.
// hipcc --offload-arch=gfx950 -O3 tools/xcd_partition_bench.hip -o /tmp/xcd_partition_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

__global__ __launch_bounds__(256) void k(const float2* __restrict__ table, uint32_t n_lines, uint32_t G, int iters, float* out) {
    const uint32_t lane = threadIdx.x & 63, gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t slice_lines = n_lines / G, slice = (blockIdx.x & 7u) * G / 8u;
    const float2* base = table + (size_t)slice * slice_lines * 8u;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        float2 v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint32_t line = mix(gw * 7919u + it * 104729u + c * 31u + (lane & 31u) * 2654435761u) % slice_lines;
            v[c] = base[(size_t)line * 8u + (lane >> 5)];
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) acc += v[c].x + v[c].y;
    }
    if (acc == 12345.f) out[0] = acc;
}

int main() {
    float2* t; float* out;
    if (hipMalloc(&t, (size_t)64u << 20) != hipSuccess || hipMemset(t, 0, (size_t)64u << 20) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) return 1;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 256, blocks = 2048;
    auto run = [&](uint32_t lines, uint32_t G, const char* tag, double mb) {
        auto launch = [&] { hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, t, lines, G, iters, out); };
        launch();
        (void)hipEventRecord(e0);
        for (int r = 0; r < 3; ++r) launch();
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
        const double lines_touched = (double)blocks * 4.0 * iters * 8.0 * 32.0;
        printf("region %5.2f MB  %-8s G=%u: %7.3f ms  %6.1f G lines/s\n", mb, tag, G, ms, lines_touched / (ms * 1e-3) / 1e9);
    };
    for (double mb : {0.5, 1.0, 1.5, 2.0, 2.5, 3.0, 3.5, 4.0, 5.0, 6.0, 6.5, 8.0}) run((uint32_t)(mb * 1024 * 1024 / 64), 1u, "whole", mb);
    for (double mb : {6.0, 6.5, 8.0}) for (uint32_t G : {2u, 4u, 8u}) run((uint32_t)(mb * 1024 * 1024 / 64), G, "part", mb);
    return 0;
}
