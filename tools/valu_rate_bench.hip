// Microbenchmark: issue cost of the fp64 / integer-multiply VALU forms next to v_fma_f32, one wave per SIMD, 8 independent chains.
// hipcc --offload-arch=gfx950 -O3 tools/valu_rate_bench.hip -o /tmp/vrb && /tmp/vrb
#include <hip/hip_runtime.h>
#include <cstdio>

// op: 0 v_fma_f32, 1 v_fma_f64, 2 v_mul_f64, 3 v_add_f64, 4 v_cvt_f64_f32 (+ v_cvt_f32_f64 to close the chain), 5 v_mul_lo_u32, 6 v_mul_u32_u24
template <int OP>
__global__ __launch_bounds__(256) void k(int iters, float* out) {
    const long long t0 = clock64();
    float f[8];
    double d[8];
    unsigned u[8];
    for (int i = 0; i < 8; ++i) { f[i] = threadIdx.x * 1e-3f + i; d[i] = f[i]; u[i] = threadIdx.x + i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
                if constexpr (OP == 1) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
                if constexpr (OP == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
                if constexpr (OP == 3) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
                if constexpr (OP == 4) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
                if constexpr (OP == 5) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                if constexpr (OP == 6) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
            }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += f[i] + (float)d[i] + (float)u[i];
    if (s == 12345.678f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[1] = (float)(clock64() - t0) / (32.0f * iters);
}

template <int OP>
void run(const char* name, float* out) {
    const int iters = 20000;
    float h[2];
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256), 0, 0, iters, out);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256), 0, 0, iters, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(h, out, 8, hipMemcpyDeviceToHost);
    // one wave per SIMD: time per instruction = issue cost; clock64 counts at 100 MHz on this part, so report nanoseconds
    printf("%-16s %.3f ns per wave instruction (%.3f ms)\n", name, ms * 1e6 / (32.0 * iters), ms);
}

int main() {
    float* out; (void)hipMalloc(&out, 64);
    run<0>("v_fma_f32", out); run<1>("v_fma_f64", out); run<2>("v_mul_f64", out); run<3>("v_add_f64", out);
    run<4>("v_cvt_f64_f32", out); run<5>("v_mul_lo_u32", out); run<6>("v_mul_u32_u24", out);
    return 0;
}
