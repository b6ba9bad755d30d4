// Microbenchmark: what the memory system of an MI355X delivers for RANDOM 64-byte lines out of a table that does not fit any
// cache (the T = 2^22 hash levels: 281 MB) -- the ceiling the hash gather of BASELINE configs[4] can reach, next to a streaming copy.
//   pattern A: every lane of a load instruction its own random line                      (64 lines / instruction)
//   pattern B: lanes l and l + 32 share a line (x-neighbour corners in the two halves)   (32 lines / instruction: the forward's layout)
//   G = independent load instructions in flight per wave (8 = one level's corners, 16 / 32 = two / four levels)
//   W = waves per SIMD the launch allows
// hipcc --offload-arch=gfx950 -O3 tools/hbm_random_line_bench.hip -o /tmp/hrl && /tmp/hrl
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int PAIR, int G, int W>
__global__ __launch_bounds__(256, W) void k(const float2* __restrict__ table, uint32_t n_lines, int iters, float* out) {
    const uint32_t lane = threadIdx.x & 63, gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        float2 v[G];
#pragma unroll
        for (int c = 0; c < G; ++c) {
            const uint32_t grp = PAIR ? (lane & 31) : lane, sub = PAIR ? lane >> 5 : 0;
            const uint32_t line = mix(gw * 7919u + it * 104729u + c * 31u + grp * 2654435761u) % n_lines;
            v[c] = table[(size_t)line * 8u + sub];
        }
#pragma unroll
        for (int c = 0; c < G; ++c) acc += v[c].x + v[c].y;
    }
    if (acc == 12345.f) out[0] = acc;
}

__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

template <int PAIR, int G, int W>
void run(const float2* t, uint32_t n_lines, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * W, iters = 4096 / G;
    hipLaunchKernelGGL((k<PAIR, G, W>), dim3(blocks), dim3(256), 0, 0, t, n_lines, iters, out);
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<PAIR, G, W>), dim3(blocks), dim3(256), 0, 0, t, n_lines, iters, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr = 3.0 * blocks * 4.0 * iters * G, lines = instr * (PAIR ? 32.0 : 64.0);
    printf("  %s  G=%2d  W=%d : %8.3f ms  %7.1f G lines/s  = %6.2f TB/s of 64-byte lines  (%5.1f G lane-gathers/s)\n", PAIR ? "pairs (32 lines/instr)" : "single (64 lines/instr)", G, W,
           ms / 3, lines / (ms * 1e-3) / 1e9, lines * 64.0 / (ms * 1e-3) / 1e12, instr * 64.0 / (ms * 1e-3) / 1e9);
}

int main() {
    float2* t; float* out; float4* dst;
    const size_t bytes = (size_t)512u << 20;
    CK(hipMalloc(&t, bytes)); CK(hipMemset(t, 0, bytes)); CK(hipMalloc(&out, 4)); CK(hipMalloc(&dst, bytes));
    {   // streaming copy: the practical HBM ceiling
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k_copy, dim3(256 * 16), dim3(256), 0, 0, (const float4*)t, dst, bytes / 16);
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_copy, dim3(256 * 16), dim3(256), 0, 0, (const float4*)t, dst, bytes / 16);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("streaming copy of 512 MB: %.3f ms  = %.2f TB/s read + write\n", ms / 5, 5.0 * 2.0 * bytes / (ms * 1e-3) / 1e12);
    }
    for (double mb : {281.0, 512.0, 32.0}) {
        const uint32_t lines = (uint32_t)(mb * 1024 * 1024 / 64);
        printf("random lines out of %.0f MB:\n", mb);
        run<0, 8, 8>(t, lines, out); run<0, 16, 8>(t, lines, out); run<0, 32, 4>(t, lines, out);
        run<1, 8, 8>(t, lines, out); run<1, 16, 8>(t, lines, out); run<1, 32, 4>(t, lines, out);
        run<1, 8, 2>(t, lines, out); run<1, 16, 2>(t, lines, out); run<1, 32, 2>(t, lines, out); run<1, 8, 4>(t, lines, out);
    }
    return 0;
}
