// Microbenchmark: do fp32 MFMA (v_mfma_f32_32x32x2_f32, 16 passes) and ordinary VALU work overlap on one SIMD
//   (a) when they come from two different waves of the SIMD, (b) when one wave interleaves them?
// hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_overlap_bench.hip -o /tmp/mvo && /tmp/mvo
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#ifdef USE_BF16      // the same experiment with v_mfma_f32_32x32x16_bf16 (8 passes) in place of the fp32 form
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, pb, c, 0, 0, 0)
#else
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)
#endif

// mode bit 0: MFMA waves active, bit 1: VALU waves active, bit 2: same-wave interleave (all 4 waves do both)
__global__ __launch_bounds__(512) void k(int mode, int iters, float* out) {
    const int wave = threadIdx.x >> 6;
    const long long t0 = clock64();
    f32x16 acc0 = {0}, acc1 = {0};
    float a = threadIdx.x * 1e-3f, b = 1.0f + a;
    bf16x8 pa, pb;
    for (int i = 0; i < 8; ++i) { pa[i] = (__bf16)(a + i); pb[i] = (__bf16)(b - i); }
    float v0 = a, v1 = b, v2 = a + 1, v3 = b + 1, v4 = a + 2, v5 = b + 2, v6 = a + 3, v7 = b + 3;
    if (mode & 4) {
        if (wave < 4) {
            for (int i = 0; i < iters; ++i) {
                acc0 = MFMA(a, b, acc0);
#pragma unroll
                for (int u = 0; u < 2; ++u) { v0 = fmaf(v0, 1.0001f, 0.5f); v1 = fmaf(v1, 1.0001f, 0.5f); v2 = fmaf(v2, 1.0001f, 0.5f); v3 = fmaf(v3, 1.0001f, 0.5f);
                                              v4 = fmaf(v4, 1.0001f, 0.5f); v5 = fmaf(v5, 1.0001f, 0.5f); v6 = fmaf(v6, 1.0001f, 0.5f); v7 = fmaf(v7, 1.0001f, 0.5f); }
                acc1 = MFMA(a, b, acc1);
#pragma unroll
                for (int u = 0; u < 2; ++u) { v0 = fmaf(v0, 1.0001f, 0.5f); v1 = fmaf(v1, 1.0001f, 0.5f); v2 = fmaf(v2, 1.0001f, 0.5f); v3 = fmaf(v3, 1.0001f, 0.5f);
                                              v4 = fmaf(v4, 1.0001f, 0.5f); v5 = fmaf(v5, 1.0001f, 0.5f); v6 = fmaf(v6, 1.0001f, 0.5f); v7 = fmaf(v7, 1.0001f, 0.5f); }
            }
        }
    } else if (wave < 4) {
        if (mode & 1)
            for (int i = 0; i < iters; ++i) {
                acc0 = MFMA(a, b, acc0);
                acc1 = MFMA(a, b, acc1);
            }
    } else {
        if (mode & 2)
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 4; ++u) { v0 = fmaf(v0, 1.0001f, 0.5f); v1 = fmaf(v1, 1.0001f, 0.5f); v2 = fmaf(v2, 1.0001f, 0.5f); v3 = fmaf(v3, 1.0001f, 0.5f);
                                              v4 = fmaf(v4, 1.0001f, 0.5f); v5 = fmaf(v5, 1.0001f, 0.5f); v6 = fmaf(v6, 1.0001f, 0.5f); v7 = fmaf(v7, 1.0001f, 0.5f); }
            }
    }
    float s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    if (s == 12345.678f) out[0] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) out[1 + wave] = (float)(clock64() - t0) / iters;      // shader-clock cycles per iteration
}

int main() {
    float* out; (void)hipMalloc(&out, 64); float h[16];
    const int iters = 20000;
    const char* names[] = {"", "MFMA waves only (2 x 64 cycles / iter)", "VALU waves only (32 fma = 128 cycles / iter)", "both, different waves of the SIMD", "", "same wave interleaves 2 MFMA + 32 fma"};
    for (int mode : {1, 2, 3, 5}) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, iters, out);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, iters, out);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
        printf("%-48s %.3f ms; shader cycles / iteration: MFMA wave %.1f, VALU wave %.1f  (clock %.2f GHz)\n", names[mode], ms, h[1], h[5],
               (h[1] > h[5] ? h[1] : h[5]) * iters / (ms * 1e-3) / 1e9);
    }
    return 0;
}
