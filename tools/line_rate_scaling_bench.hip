// Microbenchmark: is the rate at which a CU takes random 64-byte lines (8-byte gathers, lanes l and l + 32 share a line: the forward's
// x-pair layout) a PER-CU limit or a CHIP-LEVEL one (L2 / fabric)?  The same kernel is launched on 16 ... 256 CUs' worth of workgroups
// (W workgroups of 4 waves per CU): if lines per cycle and CU stay put as the active CUs shrink, the limit sits in the CU's own memory
// path; if they rise, it is shared.  Region sizes: 0.5 MB (L2-resident everywhere), 6 MB (the shipped table), 64 MB.
// hipcc --offload-arch=gfx950 -O3 tools/line_rate_scaling_bench.hip -o tools/line_rate_scaling_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int PAIR>
__global__ __launch_bounds__(256) void k(const float2* __restrict__ table, uint32_t n_lines, int iters, float* out) {
    const uint32_t lane = threadIdx.x & 63, gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        float2 v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint32_t key = PAIR ? (lane & 31u) : lane;
            const uint32_t line = mix(gw * 7919u + it * 104729u + c * 31u + key * 2654435761u) % n_lines;
            v[c] = table[(size_t)line * 8u + (PAIR ? (lane >> 5) : 0u)];
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
    const int iters = 512;
    for (uint32_t mb10 : {5u, 60u, 640u}) {
        const uint32_t lines = (uint32_t)((uint64_t)mb10 * 1024u * 1024u / 10u / 64u);
        for (int pair = 1; pair >= 0; --pair) {
            for (int W : {2, 8}) {
                for (int cus : {16, 32, 64, 128, 256}) {
                    const int blocks = cus * W;
                    auto launch = [&] { if (pair) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, t, lines, iters, out); else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, t, lines, iters, out); };
                    launch();
                    (void)hipEventRecord(e0);
                    for (int r = 0; r < 3; ++r) launch();
                    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
                    const double instr = (double)blocks * 4.0 * iters * 8.0, lines_touched = instr * (pair ? 32.0 : 64.0);
                    // blocks are spread round-robin over the XCDs, so `cus` workgroup-sets do not necessarily sit on `cus` distinct CUs; the chip-wide rate is what matters
                    printf("region %5.1f MB  %s  W=%d  workgroups %4d (%3d CUs' worth): %7.3f ms  %6.1f G lines/s  %5.1f G instr/s\n", mb10 / 10.0, pair ? "pairs " : "single", W, blocks,
                           cus, ms, lines_touched / (ms * 1e-3) / 1e9, instr / (ms * 1e-3) / 1e9);
                }
            }
        }
    }
    return 0;
}
