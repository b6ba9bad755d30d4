// Microbenchmark: do 8-byte gathers from an L2-resident table and VALU work of the SAME wave overlap on a CU, or do their costs add?
// Per iteration a wave issues 8 independent gathers (lanes l and l + 32 share a 64-byte line: the forward's x-pair layout, ~32 lines per
// instruction) and NV fused multiply-adds on private registers that do not depend on the loaded values; the loaded values are consumed
// only after the FMAs.  Variants: gathers only, FMAs only, both.  W = waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/gather_valu_overlap_bench.hip -o /tmp/gvo && /tmp/gvo
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int GATHER, int NV, int W>
__global__ __launch_bounds__(256, W) void k(const float2* __restrict__ table, uint32_t n_lines, int iters, float* out) {
    const uint32_t lane = threadIdx.x & 63, gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    float acc = 0.f;
    float r[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) r[q] = 1.0f + 0.001f * (float)(lane + q);
    uint32_t h = gw * 7919u + lane;
    for (int it = 0; it < iters; ++it) {
        float2 v[8];
        if (GATHER) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const uint32_t line = mix(gw * 7919u + it * 104729u + c * 31u + (lane & 31u) * 2654435761u) % n_lines;
                v[c] = table[(size_t)line * 8u + (lane >> 5)];
            }
        }
#pragma unroll
        for (int i = 0; i < NV / 8; ++i) {
#pragma unroll
            for (int q = 0; q < 8; ++q) r[q] = fmaf(r[q], 1.000001f, 0.5f);
        }
        if (GATHER) {
#pragma unroll
            for (int c = 0; c < 8; ++c) acc += v[c].x + v[c].y;
        }
        h = mix(h);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) acc += r[q];
    if (acc == 12345.f) out[0] = acc + (float)h;
}

template <int GATHER, int NV, int W>
float run(const float2* t, uint32_t n_lines, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * W, iters = 256;
    hipLaunchKernelGGL((k<GATHER, NV, W>), dim3(blocks), dim3(256), 0, 0, t, n_lines, iters, out);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<GATHER, NV, W>), dim3(blocks), dim3(256), 0, 0, t, n_lines, iters, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}

template <int NV, int W>
void trio(const float2* t, uint32_t lines, float* out) {
    const float g = run<1, 0, W>(t, lines, out), v = run<0, NV, W>(t, lines, out), b = run<1, NV, W>(t, lines, out);
    // per CU and iteration-of-one-wave: cycles at 2.1 GHz / (W * 4 waves per CU * 256 iterations)
    const double cyc = 2.1e6 / (W * 4.0 * 256.0);
    printf("  W=%d  %3d FMAs per 8 gathers: gathers %.3f ms (%5.0f CU-cycles per 8)  FMAs %.3f ms (%5.0f)  both %.3f ms (%5.0f)   sum %.3f  max %.3f  -> overlap %.0f %%\n", W, NV, g, g * cyc,
           v, v * cyc, b, b * cyc, g + v, g > v ? g : v, 100.0 * (g + v - b) / (g < v ? g : v));
}

int main() {
    float2* t; float* out;
    CK(hipMalloc(&t, (size_t)64u << 20)); CK(hipMemset(t, 0, (size_t)64u << 20)); CK(hipMalloc(&out, 4));
    const uint32_t lines = 6u * 1024u * 1024u / 64u;      // ~ the 6.5 MB table
    trio<64, 2>(t, lines, out); trio<128, 2>(t, lines, out); trio<256, 2>(t, lines, out); trio<512, 2>(t, lines, out);
    trio<128, 4>(t, lines, out); trio<256, 4>(t, lines, out);
    trio<128, 8>(t, lines, out); trio<256, 8>(t, lines, out);
    trio<128, 1>(t, lines, out); trio<256, 1>(t, lines, out);
    return 0;
}
