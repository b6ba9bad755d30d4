// Microbenchmark: cost of 8-byte gathers from an L2-resident table as a function of how lanes share 64-byte lines.
//   mode 0: every lane its own random line            (what lane = sample gives on the fine hash levels)
//   mode 1: lanes (2j, 2j+1) share a line             (x-neighbour corners in adjacent lanes)
//   mode 2: lanes l and l+32 share a line             (x-neighbour corners in the two wave halves)
//   mode 3: quads of 4 lanes share a line
// hipcc --offload-arch=gfx950 -O3 tools/gather_coalesce_bench.hip -o /tmp/gcb && /tmp/gcb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE>
__global__ __launch_bounds__(256) void k(const float2* __restrict__ table, uint32_t n_lines, int iters, float* out) {
    const uint32_t lane = threadIdx.x & 63, gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        float2 v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            uint32_t grp = MODE == 0 ? lane : (MODE == 1 ? lane >> 1 : (MODE == 2 ? (lane & 31) : lane >> 2));
            uint32_t sub = MODE == 0 ? 0 : (MODE == 1 ? (lane & 1) : (MODE == 2 ? lane >> 5 : (lane & 3)));
            const uint32_t line = mix(gw * 7919u + it * 104729u + c * 31u + grp * 2654435761u) % n_lines;
            v[c] = table[(size_t)line * 8u + sub];
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) acc += v[c].x + v[c].y;
    }
    if (acc == 12345.f) out[0] = acc;
}

template <int MODE>
double run(const float2* t, uint32_t n_lines, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8, iters = 64;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, t, n_lines, iters, out);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, t, n_lines, iters, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double gathers = 5.0 * blocks * 256.0 * iters * 8.0;
    printf("mode %d: %.3f ms  %.1f G lane-gathers/s\n", MODE, ms / 5, gathers / (ms * 1e-3) / 1e9);
    return ms;
}

int main() {
    const uint32_t lines = 6u * 1024u * 1024u / 64u;                  // ~ the 6.5 MB table
    float2* t; float* out;
    CK(hipMalloc(&t, (size_t)256u << 20)); CK(hipMemset(t, 0, (size_t)256u << 20)); CK(hipMalloc(&out, 4));
    run<0>(t, lines, out); run<1>(t, lines, out); run<2>(t, lines, out); run<3>(t, lines, out);
    // the layout in use (mode 2) against the size of the region the gathers fall in: L1 / L2 (4 MB per XCD) / beyond
    const double mb[] = {0.0625, 0.25, 0.5, 1, 2, 3, 4, 6.5, 16, 64, 256};
    for (double m : mb) {
        printf("region %.4g MB: ", m);
        run<2>(t, (uint32_t)(m * 1024 * 1024 / 64), out);
    }
    return 0;
}
